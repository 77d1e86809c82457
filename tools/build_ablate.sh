#!/usr/bin/env bash
# Timing-only diagnostic builds (wrong results by construction) into approximategps.jl_amd/csrc/ablate/:
#   build_ablate.sh strip N...   -DSVGP_ABLATE=N on strip.hip: bit 1 no P-tile loads, 2 no Q-tile loads, 4 no scratch stores, ...
#   build_ablate.sh stripstamps x -DSVGP_STRIP_STAMPS: clock64() stamps inside one steady-state strip (tools/strip_stamps.py)
#   build_ablate.sh stamps x     -DSVGP_POTF2_STAMPS: clock64() stamps inside potf2 (read with tools/potf2_time.py)
#   build_ablate.sh potf2 N...   -DSVGP_POTF2_ABLATE=N on prep.hip: bit 1 no 16x16 register factor, 2 no panel solve,
#                                4 no trailing update, 8 no blocked inverse
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/approximategps.jl_amd/csrc"; OUT="$SRC/ablate"; mkdir -p "$OUT"
what=$1; shift
for v in "$@"; do
  if [ "$what" = strip ]; then
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSVGP_ABLATE=$v -c "$SRC/strip.hip" -o "$OUT/strip_$v.o" 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_strip_$v.so" "$SRC/prep.o" "$OUT/strip_$v.o" "$SRC/grad.o" "$SRC/api.o" "$SRC/comm.o" -ldl
  elif [ "$what" = stripstamps ]; then
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSVGP_STRIP_STAMPS -c "$SRC/strip.hip" -o "$OUT/strip_stamps.o" 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_stripstamps.so" "$SRC/prep.o" "$OUT/strip_stamps.o" "$SRC/grad.o" "$SRC/api.o" "$SRC/comm.o" -ldl
  elif [ "$what" = stamps ]; then
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSVGP_POTF2_STAMPS -c "$SRC/prep.hip" -o "$OUT/prep_stamps.o" 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_stamps.so" "$OUT/prep_stamps.o" "$SRC/strip.o" "$SRC/grad.o" "$SRC/api.o" "$SRC/comm.o" -ldl
  else
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSVGP_POTF2_ABLATE=$v -c "$SRC/prep.hip" -o "$OUT/prep_$v.o" 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgp_potf2_$v.so" "$OUT/prep_$v.o" "$SRC/strip.o" "$SRC/grad.o" "$SRC/api.o" "$SRC/comm.o" -ldl
  fi
done
ls "$OUT"/*.so
