#!/bin/bash
# round 3, first GPU visit: full GPU suite, the default bench line, the same with the RCCL path forced on one GPU, small-problem table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r3/pytest.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_H.json 2> gpurun_out/r3/bench_H.err
BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kuf > gpurun_out/r3/bench_H_forcedist.json 2> gpurun_out/r3/bench_H_forcedist.err
timeout 900 python tools/small_time.py > gpurun_out/r3/small_time.log 2>&1
tail -5 gpurun_out/r3/pytest.log
python - <<'PY'
import json
for f in ("bench_H", "bench_H_forcedist"):
    try:
        d = json.loads(open(f"gpurun_out/r3/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("rccl_world"), d.get("parity"), d.get("value_and_gradient"), {k: v for k, v in d.get("c5_minibatch", {}).items() if "training" in k or k == "ms_per_step"}, d.get("cpu_baseline", {}).get("julia"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/r3/small_time.log
