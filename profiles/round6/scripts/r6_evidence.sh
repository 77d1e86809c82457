#!/bin/bash
# round-6 evidence run (one gpurun call): GPU suite on the product and the experiments library, rocprofv3 profiles (stats + PMC passes)
# of H / H grad / C2 / C4 / C5 / C5 grad / Hd64 grad, one bench line per configuration, the default bench line, the build log.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_prof; mkdir -p $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | f | tail -n 3) > $O/gputest_product.log 2>&1; cat $O/gputest_product.log
(SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so timeout 1700 python -m pytest tests -m gpu -q 2>&1 | f | tail -n 3) > $O/gputest_experiments.log 2>&1; cat $O/gputest_experiments.log
for C in H C2 C4 C5; do bash tools/run_profile.sh r6_${C} $C > $O/prof_${C}.log 2>&1; tail -1 $O/prof_${C}.log | cut -c1-160; done
bash tools/run_profile.sh r6_Hgrad H grad > $O/prof_Hgrad.log 2>&1
bash tools/run_profile.sh r6_C5grad C5 grad > $O/prof_C5grad.log 2>&1
bash tools/run_profile.sh r6_Hd64grad Hd64 grad > $O/prof_Hd64grad.log 2>&1
{
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('value_and_gradient',{}); b=d['breakdown_ms']; k=d.get('kuf_roofline',{}); r=d['roofline']; print(sys.argv[1], round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms | roofline', r['kernel'].split(' ')[0], 'TF', round(r['achieved'],1), 'frac', round(r['frac'],3), '| kuf GB/s', round(k.get('achieved') or 0), 'p95', round(k.get('GBps_p95_launch',0)), 'fill', round(k.get('stream_write_GBps') or 0), '| grad ms', round(g.get('ms_per_eval',0),2), 'x', round(g.get('ratio_to_forward',0),2), '| chol ms', round(d['cholesky_roofline']['ms'],3), 'frac', round(d['cholesky_roofline']['frac'] or 0,4), '|', {k.split(' ')[0]: round(v,3) for k,v in b.items()})" "$1"; }
for c in H H32 C2 C3 C4 C5 Hd16 Hd17 Hd24 Hd32 Hd48 Hd64 H32d32 H32d64 MB16k MB4k; do timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-c5 2>/dev/null | pr $c; done
} 2>&1 | tee $O/all_configs_r6.log
timeout 900 python bench.py > $O/bench_H_final.json 2> $O/bench_H_final.err; cut -c1-300 $O/bench_H_final.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | f | tail -n 3 | tee $O/smoke.log
# the driver's N > 1 launch form, on this one-GPU box with one rank
(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-c5 2>&1 | f | tail -n 1 | cut -c1-300) | tee $O/bench_torchrun_n1.log
cp approximategps.jl_amd/csrc/build.log $O/build.log 2>/dev/null
# soak runs on the final library (create / evaluate / free loop; strips beside / behind the factorisation)
(timeout 200 python tests/soak.py 60 2>&1 | tail -n 3; timeout 300 python tests/soak_overlap.py 200 2>&1 | tail -n 2) | tee $O/soak.log
