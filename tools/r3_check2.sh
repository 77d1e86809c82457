#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/r3/pytest_full.log 2>&1
head -40 gpurun_out/r3/pytest_full.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python tools/chol_check.py | tail -1
for c in C4 C3; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-kuf --no-grad --no-c5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['breakdown_ms'], d['cholesky_roofline']['frac'])"; done
python tools/small_time.py > gpurun_out/r3/small_time.log 2>&1; tail -2 gpurun_out/r3/small_time.log
