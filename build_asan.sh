#!/usr/bin/env bash
# AddressSanitizer build of the HOST side of libsvgp_mi355x: the device code objects are compiled as usual
# (-fno-gpu-sanitize: GPU ASan / xnack+ code objects are not available on the target pool), every host function - the
# C-ABI, the Golub-Welsch rule, handle management, the RCCL loader - is instrumented.  Used by tests/test_asan_cpu.py,
# which runs the no-GPU part of the ABI tests against it in a child process with the ASan runtime preloaded.
# CPU only: never load this build on the GPU box.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/approximategps.jl_amd/csrc"; OUT="$SRC/asan"; mkdir -p "$OUT"
FLAGS="-O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-function"
pids=()
for f in prep strip grad api comm; do
  hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o "$OUT/libsvgp_mi355x_asan.so" "$OUT"/{prep,strip,grad,api,comm}.o -ldl
echo "built $OUT/libsvgp_mi355x_asan.so"
