#!/usr/bin/env bash
# Builds the library of git HEAD into csrc/ablate/libsvgp_prev.so for same-box A/B timing against the working tree.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; TMP=$(mktemp -d); OUT="$ROOT/approximategps.jl_amd/csrc/ablate"; mkdir -p "$OUT"
git -C "$ROOT" archive HEAD approximategps.jl_amd/csrc include | tar -x -C "$TMP"
for f in prep strip grad api comm; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -c "$TMP/approximategps.jl_amd/csrc/$f.hip" -o "$TMP/$f.o" 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_prev.so" "$TMP/prep.o" "$TMP/strip.o" "$TMP/grad.o" "$TMP/api.o" "$TMP/comm.o" -ldl
rm -rf "$TMP"; echo "built $OUT/libsvgp_prev.so from $(git -C "$ROOT" rev-parse --short HEAD)"
