#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_prof; mkdir -p $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | f | tail -n 3) > $O/gputest_product.log 2>&1; cat $O/gputest_product.log
(SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so timeout 1700 python -m pytest tests -m gpu -q 2>&1 | f | tail -n 3) > $O/gputest_experiments.log 2>&1; cat $O/gputest_experiments.log
