cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_21; mkdir -p $O
A=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prio.so
for rep in 1 2; do for c in C5 C2 H C4; do python tools/ablate_time.py $c 2>/dev/null | sed "s/^/base /"; SVGP_MI355X_LIB=$A python tools/ablate_time.py $c 2>/dev/null | sed "s/^/prio /"; done; done | tee $O/prio_fwd.log
for rep in 1 2; do for c in C5 H H32; do python tools/grad_time.py $c 2>/dev/null | grep elbo_grad | sed "s/^/base /"; SVGP_MI355X_LIB=$A python tools/grad_time.py $c 2>/dev/null | grep elbo_grad | sed "s/^/prio /"; done; done | tee $O/prio_grad.log
for cc in 131072 262144; do SVGP_GRAD_CHUNK=$cc SVGP_GRAD_CHUNK_BYTES=4e9 python tools/grad_time.py H 2>/dev/null | grep elbo_grad | sed "s/^/chunk=$cc /"; done | tee $O/chunk.log
