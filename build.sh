#!/usr/bin/env bash
# Builds libsvgp_mi355x.so for gfx950 in-tree (approximategps.jl_amd/csrc/).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/approximategps.jl_amd/csrc"
OUT="$SRC/libsvgp_mi355x.so"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function ${SVGP_EXTRA_FLAGS:-}"
pids=()
for f in prep strip grad api comm; do
  hipcc $FLAGS -c "$SRC/$f.hip" -o "$SRC/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$SRC/prep.o" "$SRC/strip.o" "$SRC/grad.o" "$SRC/api.o" "$SRC/comm.o" -ldl
echo "built $OUT"
