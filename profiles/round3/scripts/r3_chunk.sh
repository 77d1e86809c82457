#!/bin/bash
# gradient chunk size sweep (knobs SVGP_GRAD_CHUNK points, SVGP_GRAD_CHUNK_BYTES per operand buffer)
mkdir -p gpurun_out/r3
for cfg in C2 H C5 C3 H32; do
  for c in 32768 65536 131072 262144; do
    echo "chunk=$c: $(SVGP_GRAD_CHUNK=$c SVGP_GRAD_CHUNK_BYTES=8e9 python3 tools/grad_time.py $cfg | tail -1)"
  done
done 2>&1 | tee gpurun_out/r3/grad_chunk.log
