#!/usr/bin/env bash
# Builds the microbenchmarks of this directory (and tests/probe_lds) for gfx950; the executables are git-ignored.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
for f in mfma_direct mfma_f64 mfma_struct valu_f64 store_pattern dpp_bcast clock_rate; do
  hipcc -O3 -std=c++17 -Wno-unused-value --offload-arch=gfx950 "$HERE/$f.hip" -o "$HERE/$f"
done
hipcc -O3 -std=c++17 --offload-arch=gfx950 "$HERE/../../tests/probe_lds.hip" -o "$HERE/../../tests/probe_lds"
