#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -n 5
