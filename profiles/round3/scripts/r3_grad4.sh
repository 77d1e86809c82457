#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r3/pytest_grad4.log 2>&1
head -60 gpurun_out/r3/pytest_grad4.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
