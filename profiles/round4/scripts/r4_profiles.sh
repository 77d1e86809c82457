#!/bin/bash
# round-4 evidence run: GPU suite, profiles (stats + PMC passes) of H / H grad / C2 / C4 / C5 / C5 grad, one bench line per config
# (forward + value-and-gradient, incl. the wide-input shapes), the sustained leg, the small-problem and overlap tables
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_prof; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/pytest_final.log 2>&1; cat $O/pytest_final.log
timeout 600 python bench.py --steps 20 --warmup 5 --min-seconds 30 --sustained-out $O/sustained_H.json --no-c5 --no-grad --no-cpu-baseline --no-kuf > $O/bench_sustained.json 2> $O/bench_sustained.err
for C in H C2 C4 C5; do bash tools/run_profile.sh r4_${C} $C > $O/prof_${C}.log 2>&1; tail -1 $O/prof_${C}.log | cut -c1-160; done
bash tools/run_profile.sh r4_Hgrad H grad > $O/prof_Hgrad.log 2>&1
bash tools/run_profile.sh r4_C5grad C5 grad > $O/prof_C5grad.log 2>&1
{
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('value_and_gradient',{}); b=d['breakdown_ms']; k=d.get('kuf_roofline',{}); print(sys.argv[1], round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms | strip TF', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],3), '| kuf GB/s', round(k.get('achieved',0)), 'p95', round(k.get('GBps_p95_launch',0)), 'fill', round(k.get('stream_write_GBps') or 0), '| grad ms', round(g.get('ms_per_eval',0),2), 'x', round(g.get('ratio_to_forward',0),2), '|', {k.split(' ')[0]: round(v,3) for k,v in b.items()})" "$1"; }
for c in H H32 C2 C3 C4 C5 Hd17 Hd32 Hd64 H32d32 H32d64 MB16k MB4k; do timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-c5 2>/dev/null | pr $c; done
} 2>&1 | tee $O/all_configs_r4.log
timeout 600 python tools/overlap_time.py f64 2>&1 | grep "n=" | tee $O/overlap_f64.log
timeout 600 python tools/overlap_time.py f32 2>&1 | grep "n=" | tee $O/overlap_f32.log
timeout 600 python tools/overlap_grad_time.py f64 2>&1 | grep "n=" | tee $O/overlap_grad_f64.log
timeout 600 python tools/overlap_grad_time.py f32 2>&1 | grep "n=" | tee $O/overlap_grad_f32.log
timeout 900 python tests/small_time.py > $O/small_time.log 2>&1; cp gpurun_out/small_problems.md $O/ 2>/dev/null; tail -3 $O/small_time.log
timeout 900 python bench.py > $O/bench_H_final.json 2> $O/bench_H_final.err; cut -c1-300 $O/bench_H_final.json
