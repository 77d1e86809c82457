cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kuf or golden" 2>&1 | tail -3)
for c in H C2 H32 C3; do timeout 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'kuf GB/s', round(d['kuf_roofline']['achieved']), 'kuf ms', round(d['kuf_roofline']['ms_per_launch'],3))" $c; done
