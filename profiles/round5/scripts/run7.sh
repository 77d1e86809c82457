#!/usr/bin/env bash
# round 5: two-stream look-ahead of the large-Kuu Cholesky: tests, A/B (experiments library), race check, C4 bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_round5.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/r5/gputest_c.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5/gputest_c.log | tail -n 4
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so python tools/round5/chol_lookahead_ab.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r5/chol_lookahead_ab.log
python tools/chol_check.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -n 12 | tee gpurun_out/r5/chol_check_final.log
python bench.py --config C4 --steps 20 --warmup 3 --no-cpu-baseline --no-c5 --no-grad 2>/dev/null > gpurun_out/r5/bench_C4.json; python -c "
import json; d=json.loads(open('gpurun_out/r5/bench_C4.json').read().strip().splitlines()[-1]); print('C4', d['value'], d['ms_per_step'], d['breakdown_ms'], d['cholesky_roofline']['ms'], d['cholesky_roofline']['frac'])"
