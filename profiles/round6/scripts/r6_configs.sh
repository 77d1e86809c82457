#!/bin/bash
# one bench line per configuration + the default bench line (the second half of r6_evidence.sh, for another box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_cfg; mkdir -p $O
{
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('value_and_gradient',{}); b=d['breakdown_ms']; k=d.get('kuf_roofline',{}); r=d['roofline']; print(sys.argv[1], round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms | roofline', r['kernel'].split(' ')[0], 'TF', round(r['achieved'],1), 'frac', round(r['frac'],3), '| kuf GB/s', round(k.get('achieved') or 0), 'p95', round(k.get('GBps_p95_launch',0)), 'fill', round(k.get('stream_write_GBps') or 0), '| grad ms', round(g.get('ms_per_eval',0),2), 'x', round(g.get('ratio_to_forward',0),2), '| chol ms', round(d['cholesky_roofline']['ms'],3), 'frac', round(d['cholesky_roofline']['frac'] or 0,4), '|', {k.split(' ')[0]: round(v,3) for k,v in b.items()})" "$1"; }
for c in H H32 C2 C3 C4 C5 Hd16 Hd17 Hd24 Hd32 Hd48 Hd64 H32d32 H32d64 MB16k MB4k; do timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-c5 2>/dev/null | pr $c; done
} 2>&1 | tee $O/all_configs_r6.log
timeout 900 python bench.py > $O/bench_H_final.json 2> $O/bench_H_final.err; cut -c1-300 $O/bench_H_final.json
cp approximategps.jl_amd/csrc/build.log $O/build.log 2>/dev/null
