#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r3/pytest_chol.log 2>&1
tail -3 gpurun_out/r3/pytest_chol.log
for f in 0 1; do echo "SVGP_CHOL_FUSE=$f"; SVGP_CHOL_FUSE=$f python tools/prep_time.py; done 2>&1 | tee gpurun_out/r3/prep_ab.log
for f in 0 1; do echo "SVGP_CHOL_FUSE=$f C4"; SVGP_CHOL_FUSE=$f python bench.py --config C4 --steps 5 --warmup 2 --no-cpu-baseline --no-kuf --no-grad --no-c5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['breakdown_ms'], d['config']['elbo'])"; done 2>&1 | tee -a gpurun_out/r3/prep_ab.log
