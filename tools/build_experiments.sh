#!/usr/bin/env bash
# The EXPERIMENTS build of the library (knobs.hpp): the same sources with -DSVGP_EXPERIMENTS - every tuning knob / A-B switch read from the
# environment and the measured-and-rejected variants of rounds 1-5 compiled in - as approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so.
# Use it through SVGP_MI355X_LIB=<that path> (approxgp/_ffi.py); it exports svgp_debug_experiments, the product library does not.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/approximategps.jl_amd/csrc"; OUT="$SRC/ablate"; mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DSVGP_EXPERIMENTS ${SVGP_EXTRA_FLAGS:-}"
pids=()
for f in prep strip grad api comm; do
  hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/${f}_exp.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_experiments.so" "$OUT"/{prep,strip,grad,api,comm}_exp.o -ldl
echo "$OUT/libsvgp_experiments.so"
