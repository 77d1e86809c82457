#!/usr/bin/env bash
# the GPU parity / gradient suites of the PRODUCT library under each operational setting
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_ops; mkdir -p $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for e in "SVGP_OVERLAP=0" "SVGP_SEG_SPLIT=0" "SVGP_TIMING=0" "SVGP_DEBUG_SYNC=1"; do
  echo "== $e"
  (export $e; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py tests/test_gpu_round5.py -m gpu -q 2>&1 | f | tail -n 2)
done 2>&1 | tee $O/operational_settings_pass.log
