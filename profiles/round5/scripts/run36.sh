#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kgrad; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for rep in 1 2 3; do
for lib in default kgmw8; do
  if [ $lib = default ]; then unset SVGP_MI355X_LIB; else export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_$lib.so; fi
  for c in H C5 H32 C2; do echo -n "$lib "; timeout 900 python tools/grad_time.py $c 2>&1 | f | grep elbo_grad | cut -c1-200; done
done; done | tee $O/kgrad_minw8_ab.log
