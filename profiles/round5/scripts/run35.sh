#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_soak; mkdir -p $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
(timeout 900 python tests/soak.py 2>&1 | f | tail -n 6) | tee $O/soak.log
(timeout 900 python tests/soak_overlap.py 2>&1 | f | tail -n 6) | tee $O/soak_overlap.log
