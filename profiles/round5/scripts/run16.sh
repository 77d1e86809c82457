#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kuf; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py -k kuf_assembly -m gpu -q -x > $O/kuf_test.log 2>&1; tail -n 12 $O/kuf_test.log | cut -c1-300
for lib in default experiments; do
  if [ $lib = experiments ]; then export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so; export SVGP_KUF_COLS_WIDE=0; fi
  echo "== $lib (COLS_WIDE=${SVGP_KUF_COLS_WIDE:-1})"
  timeout 600 python tools/kuf_time.py Hd64 H32d64 Hd32 Hd17 H32d32 H 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a $O/kuf_wide.log | cut -c1-200
done
