cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/g14_tests.log 2>&1; grep -E "passed|failed|FAILED|Error" gpurun_out/g14_tests.log | head -5
for c in H C2 H32 C3; do timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'evals/s', round(d['value'],2), 'strip ms', round(d['breakdown_ms']['strip'],2), 'kuf GB/s', round(d['kuf_roofline']['achieved']), 'kuf ms', round(d['kuf_roofline']['ms_per_launch'],3))" $c; done
P=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so
for cfg in H C2; do python tools/ablate_time.py $cfg 2>/dev/null; SVGP_MI355X_LIB=$P python tools/ablate_time.py $cfg 2>/dev/null; done
