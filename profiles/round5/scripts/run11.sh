#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so timeout 900 python tools/round5/syrk_xcd_ab.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r5/syrk_xcd_ab.log
