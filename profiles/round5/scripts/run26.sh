#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kgrad; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for rep in 1 2; do
for lib in default kg8kv1; do
  if [ $lib = default ]; then unset SVGP_MI355X_LIB; else export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_$lib.so; fi
  echo "== $lib"
  for c in H C5 H32 C2 Hd16; do timeout 900 python tools/grad_time.py $c 2>&1 | f | grep elbo_grad | tee -a $O/kgrad_kv8_ab.log | cut -c1-200; done
done; done
unset SVGP_MI355X_LIB
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_round4.py -m gpu -q -x 2>&1 | f | tail -n 2
