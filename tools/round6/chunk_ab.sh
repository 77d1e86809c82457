#!/usr/bin/env bash
# value-and-gradient time against the chunk size of the point-major buffers (experiments build: SVGP_GRAD_CHUNK / SVGP_GRAD_CHUNK_BYTES)
cd $GRAFT_REPO_ROOT
export SVGP_MI355X_LIB=$GRAFT_REPO_ROOT/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so
for C in "$@"; do
  for N in ${CHUNKS:-65536 131072 262144}; do
    echo "$C chunk $N: $(SVGP_GRAD_CHUNK=$N SVGP_GRAD_CHUNK_BYTES=1e10 python3 tools/grad_time.py $C | grep elbo_grad)"
  done
done
