cd $GRAFT_REPO_ROOT
python tools/prep_time.py
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in H C2 C5 C4; do timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], round(d['value'],2), 'evals/s', round(d['ms_per_step'],3),'ms | strip frac', round(d['roofline']['frac'],3), '| prep', round(d['breakdown_ms']['prep (Kuu, cholesky, T panels, KL)'],3), '| grad ms', round(d['value_and_gradient']['ms_per_eval'],2))"; done
