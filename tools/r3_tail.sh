#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python tools/chol_check.py 2>&1 | tail -11
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_stamps.so python tools/potf2_time.py 2>&1 | tee gpurun_out/r3/potf2_stamps.log
python tools/prep_time.py 2>&1 | tee gpurun_out/r3/prep_time.log
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/r3/pytest_tail.log 2>&1
head -40 gpurun_out/r3/pytest_tail.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
bash tools/trace_eval.sh mb16k_new tools/mb_grad.py 16384 1024 8
python tools/grad_time.py C2; python tools/grad_time.py C5
python bench.py --config C4 --steps 5 --warmup 2 --no-cpu-baseline --no-kuf --no-grad --no-c5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4', d['breakdown_ms'], d['config']['elbo'], d['cholesky_roofline'])"
