#!/bin/bash
# full GPU suite + default bench + all-config lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/r3/pytest_full.log 2>&1
head -40 gpurun_out/r3/pytest_full.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python tools/chol_check.py | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_H.json 2> gpurun_out/r3/bench_H.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/bench_H.json").read().strip().splitlines()[-1])
print("H", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["breakdown_ms"], d.get("parity", {}).get("rel_err"), d.get("value_and_gradient"), {k: v for k, v in d.get("c5_minibatch", {}).items() if "training" in k or k == "ms_per_step"}, d["cholesky_roofline"])
PY
STEPS=10 bash tools/run_all.sh 2>/dev/null | tail -7
