#!/usr/bin/env bash
# Timing-only diagnostic builds of the strip kernel (wrong results by construction):
# bit 1 = no P-tile loads, bit 2 = no Q-tile loads, bit 4 = no scratch-strip stores.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/approximategps.jl_amd/csrc"; OUT="$SRC/ablate"; mkdir -p "$OUT"
for v in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSVGP_ABLATE=$v -c "$SRC/strip.hip" -o "$OUT/strip_$v.o" 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_ablate_$v.so" "$SRC/prep.o" "$OUT/strip_$v.o" "$SRC/api.o"
done
ls "$OUT"/*.so
