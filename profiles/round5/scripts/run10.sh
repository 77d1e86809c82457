#!/usr/bin/env bash
# stamps of the block factorisation INSIDE the fused small-grid launch (M = 256: the second block), 256- vs 512-thread form
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/approximategps.jl_amd/csrc/ablate
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for w in 4 8; do for dt in f64 f32; do echo "-- fused launch, SVGP_POTF2_WAVES=$w $dt"; SVGP_OVERLAP=0 POTF2_M=256 POTF2_DTYPES=$dt SVGP_POTF2_WAVES=$w SVGP_MI355X_LIB=$L/libsvgp_p_w4f.so timeout 120 python tools/potf2_time.py 2>&1 | f; done; done | tee gpurun_out/r5/potf2_fused_stamps.log | cut -c1-270
