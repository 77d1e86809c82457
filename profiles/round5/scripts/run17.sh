#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kuf; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py -k kuf_assembly -m gpu -q -x > $O/kuf_test2.log 2>&1; tail -n 3 $O/kuf_test2.log | cut -c1-300
export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so
for nb in 0 1 0 1; do
  echo "== experiments NBLK2=$nb"
  SVGP_KUF_NBLK2=$nb timeout 600 python tools/kuf_time.py Hd17 Hd24 Hd32 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a $O/kuf_nblk2.log | cut -c1-200
done
SVGP_KUF_NBLK2=1 timeout 900 python -m pytest tests/test_gpu_round5.py -k kuf_assembly -m gpu -q -x 2>&1 | tail -n 3
unset SVGP_MI355X_LIB
timeout 600 python tools/kuf_time.py Hd48 Hd64 H32d64 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a $O/kuf_nblk2.log | cut -c1-200
