#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_syrkprobe; mkdir -p $O
export TMPDIR=/tmp SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so
timeout 1200 python tools/round5/chol_env_ab.py SVGP_SYRK_PROBE 0 1 2 3 -- f32:8192 f64:4096 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $O/probe.log | cut -c1-200
