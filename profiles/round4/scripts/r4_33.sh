cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_33; mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4_33/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "stale", d["roofline"].get("traffic_stale"))
print("grad", d["value_and_gradient"]); print("host step", d.get("host_training_step"))
print("c5", {k: d["c5_minibatch"].get(k) for k in ("ms_per_step","training_step_ms")}); print("parity", d["parity"]["ok"], "lib", d["lib_sha16"])
PY
timeout 2000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tee $O/pytest.log
