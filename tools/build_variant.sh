#!/usr/bin/env bash
# A/B / bisection builds of prep.hip with extra -D flags: tools/build_variant.sh <tag> <flags...>  ->  csrc/ablate/libsvgp_<tag>.so
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/approximategps.jl_amd/csrc"; OUT="$SRC/ablate"; mkdir -p "$OUT"
tag=$1; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function "$@" -c "$SRC/prep.hip" -o "$OUT/prep_$tag.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_$tag.so" "$OUT/prep_$tag.o" "$SRC/strip.o" "$SRC/grad.o" "$SRC/api.o" "$SRC/comm.o" -ldl
echo "$OUT/libsvgp_$tag.so"
