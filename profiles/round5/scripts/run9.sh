#!/usr/bin/env bash
# serial prep (SVGP_OVERLAP=0: the chain alone on the chip) with 256- vs 512-thread fused factorisation launches (experiments library)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/approximategps.jl_amd/csrc/ablate
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for w in 4 8 4 8; do echo "-- experiments library, SVGP_OVERLAP=0, SVGP_POTF2_WAVES=$w"; SVGP_OVERLAP=0 SVGP_POTF2_WAVES=$w SVGP_MI355X_LIB=$L/libsvgp_experiments.so timeout 300 python tools/prep_time.py 2>&1 | f; done | tee gpurun_out/r5/prep_serial_w48.log
