cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_14; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; grep -n "^FAILED\|passed\|failed" $O/pytest_full.log | tail -8
{
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('value_and_gradient',{}); b=d['breakdown_ms']; k=d.get('kuf_roofline',{}); print(sys.argv[1], round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms | strip TF', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],3), '| kuf GB/s', round(k.get('achieved',0)), 'p95', round(k.get('GBps_p95_launch',0)), 'fill', round(k.get('stream_write_GBps') or 0), '| grad ms', round(g.get('ms_per_eval',0),2), 'x', round(g.get('ratio_to_forward',0),2), '|', {k.split(' ')[0]: round(v,3) for k,v in b.items()})" "$1"; }
for c in Hd17 Hd32 Hd64 H32d32 H32d64; do timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-c5 2>/dev/null | pr $c; done
} 2>&1 | tee $O/wide_configs.log
