cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_61; mkdir -p $O
for rep in 1 2; do for ch in 65536 131072 262144 32768; do
SVGP_GRAD_CHUNK=$ch SVGP_GRAD_CHUNK_BYTES=8e9 python tools/grad_time.py C5 2>/dev/null | grep grad | sed "s/^/chunk=$ch /"
SVGP_GRAD_CHUNK=$ch SVGP_GRAD_CHUNK_BYTES=8e9 python tools/grad_time.py H 2>/dev/null | grep grad | sed "s/^/chunk=$ch /"
SVGP_GRAD_CHUNK=$ch SVGP_GRAD_CHUNK_BYTES=8e9 python tools/grad_time.py H32 2>/dev/null | grep grad | sed "s/^/chunk=$ch /"
done; done | tee $O/chunk.log
