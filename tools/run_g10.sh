cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/g10_tests.log 2>&1; grep -E "passed|failed|FAILED|Error" gpurun_out/g10_tests.log | head
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],3), 'evals/s', round(d['ms_per_step'],2), 'ms strip TF', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],3), {k[:6]: round(v,3) for k,v in d['breakdown_ms'].items()})" "$1"; }
for c in H H32 C2 C3 C4 C5; do timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-kuf 2>/dev/null | pr "$c"; done
