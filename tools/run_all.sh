cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4)
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); g=d.get('value_and_gradient',{}); print(sys.argv[1], round(d['value'],2), 'evals/s', round(d['ms_per_step'],2), 'ms | strip TF', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],3), '| prep', round(d['breakdown_ms']['prep (Kuu, cholesky, T panels, KL)'],2), '| kuf GB/s', round(d['kuf_roofline']['achieved']), '| grad ms', round(g.get('ms_per_eval',0),1))" "$1"; }
for c in H H32 C2 C3 C4 C5; do timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | pr $c; done
