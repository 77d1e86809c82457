#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_cbk; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for rep in 1 2 3; do
for lib in default cbk32; do
  if [ $lib = default ]; then unset SVGP_MI355X_LIB; else export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_$lib.so; fi
  for M in 512 1024 2048 4096; do echo -n "$lib "; timeout 600 python tools/round5/chol_once.py f64 $M 2>&1 | f | tail -1; done
done; done | tee $O/chol_bk32_ab.log
export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_cbk32.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | f | tail -n 2
timeout 600 python tools/chol_check.py 2>&1 | f | tail -n 6
