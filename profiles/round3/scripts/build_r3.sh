#!/usr/bin/env bash
# Builds the round-3 library (commit 512df24) into csrc/ablate/libsvgp_r3.so for same-box A/B timing against the working tree.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; TMP=$(mktemp -d); OUT="$ROOT/approximategps.jl_amd/csrc/ablate"; mkdir -p "$OUT"
git -C "$ROOT" archive ${1:-512df24} approximategps.jl_amd/csrc include | tar -x -C "$TMP"
for f in prep strip grad api comm; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -c "$TMP/approximategps.jl_amd/csrc/$f.hip" -o "$TMP/$f.o" 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_r3.so" "$TMP/prep.o" "$TMP/strip.o" "$TMP/grad.o" "$TMP/api.o" "$TMP/comm.o" -ldl
rm -rf "$TMP"; echo "built $OUT/libsvgp_r3.so"
