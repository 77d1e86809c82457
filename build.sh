#!/usr/bin/env bash
# Builds libsvgp_mi355x.so for gfx950 in-tree (approximategps.jl_amd/csrc/).  hipcc cross-compiles without a GPU.
# Leaves approximategps.jl_amd/csrc/build.log (untracked; SVGP_BUILD_LOG overrides the path): the hipcc command lines, the compiler
# version and the sha256 of every source, object and of the library, so that a reader can see the .so a GPU box loaded (bench.py prints
# its sha256 as `lib_sha16`) is this tree's.  The evidence script of a round copies it into profiles/roundN/build.log - a plain build
# (or an ablation build with SVGP_EXTRA_FLAGS) never touches tracked files.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/approximategps.jl_amd/csrc"
OUT="$SRC/libsvgp_mi355x.so"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function ${SVGP_EXTRA_FLAGS:-}"
LOG="${SVGP_BUILD_LOG:-$SRC/build.log}"
mkdir -p "$(dirname "$LOG")"
{
  echo "# build.sh: $(hipcc --version 2>/dev/null | grep -m1 -i 'HIP version' || echo 'hipcc version unknown')"
  echo "# $(hipcc --version 2>/dev/null | grep -m1 -i 'clang version' || true)"
} > "$LOG"
pids=()
for f in prep strip grad api comm; do
  echo "hipcc $FLAGS -c approximategps.jl_amd/csrc/$f.hip -o approximategps.jl_amd/csrc/$f.o" >> "$LOG"
  hipcc $FLAGS -c "$SRC/$f.hip" -o "$SRC/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
echo "hipcc --offload-arch=gfx950 -shared -fPIC -o approximategps.jl_amd/csrc/libsvgp_mi355x.so {prep,strip,grad,api,comm}.o -ldl" >> "$LOG"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$SRC/prep.o" "$SRC/strip.o" "$SRC/grad.o" "$SRC/api.o" "$SRC/comm.o" -ldl
{
  echo "# sha256 (sources, headers, objects, library)"
  (cd "$HERE" && sha256sum approximategps.jl_amd/csrc/*.hip approximategps.jl_amd/csrc/*.hpp include/svgp_mi355x.h \
     approximategps.jl_amd/csrc/{prep,strip,grad,api,comm}.o approximategps.jl_amd/csrc/libsvgp_mi355x.so)
} >> "$LOG"
echo "built $OUT"
