#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kgrad; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_grad.py -m gpu -q -x 2>&1 | f | tail -n 3
export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so
for rep in 1 2; do
for v in 0 1 2; do
  echo "== SVGP_KGRAD_WIDE2=$v"
  for c in Hd32 Hd64 H32d32 H32d64; do SVGP_KGRAD_WIDE2=$v timeout 900 python tools/grad_time.py $c 2>&1 | f | grep elbo_grad | tee -a $O/kgrad_wide3_ab.log | cut -c1-200; done
done; done
SVGP_KGRAD_WIDE2=2 timeout 900 python -m pytest tests/test_gpu_round4.py -k wide_inputs -m gpu -q -x 2>&1 | f | tail -n 2
