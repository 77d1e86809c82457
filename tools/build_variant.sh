#!/usr/bin/env bash
# A/B / bisection builds of ONE source file with extra -D flags: [FILE=grad] tools/build_variant.sh <tag> <flags...>  ->  csrc/ablate/libsvgp_<tag>.so
# (FILE defaults to prep; the other objects are the in-tree ones: run ./build.sh first)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/approximategps.jl_amd/csrc"; OUT="$SRC/ablate"; mkdir -p "$OUT"
tag=$1; shift
F=${FILE:-prep}
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function "$@" -c "$SRC/$F.hip" -o "$OUT/${F}_$tag.o"
objs=""
for o in prep strip grad api comm; do if [ "$o" = "$F" ]; then objs="$objs $OUT/${F}_$tag.o"; else objs="$objs $SRC/$o.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_$tag.so" $objs -ldl
echo "$OUT/libsvgp_$tag.so"
