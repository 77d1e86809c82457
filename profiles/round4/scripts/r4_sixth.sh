cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_sixth; mkdir -p $O
A=$PWD/approximategps.jl_amd/csrc/ablate
for c in C5 H32 H; do for rep in 1 2; do
  python tools/grad_time.py $c 2>/dev/null | grep elbo_grad | sed "s/^/base    /"
  for v in ab32 ab64 ab128 ab224 syrknow; do SVGP_MI355X_LIB=$A/libsvgp_$v.so python tools/grad_time.py $c 2>/dev/null | grep elbo_grad | sed "s/^/$v /"; done
done; done | tee $O/grad_ablate.log
