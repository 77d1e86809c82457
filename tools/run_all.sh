#!/bin/bash
# Full GPU suite + one bench line per configuration (compact summary); output kept in gpurun_out/all.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4)
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); g=d.get('value_and_gradient',{}); b=d['breakdown_ms']; print(sys.argv[1], round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms | strip TF', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],3), '| kuf GB/s', round(d['kuf_roofline']['achieved']), '| grad ms', round(g.get('ms_per_eval',0),2), '|', {k.split(' ')[0]: round(v,3) for k,v in b.items()})" "$1"; }
for c in H H32 C2 C3 C4 C5; do timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | pr $c; done
} 2>&1 | tee gpurun_out/all.log
