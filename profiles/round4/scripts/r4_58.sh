cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_58; mkdir -p $O
for rep in 1 2; do
for c in C4 C5; do
python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-c5 --no-grad --no-kuf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', '$c', round(d['ms_per_step'],3), d['breakdown_ms'], d.get('cholesky_roofline',{}).get('frac'))"
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-c5 --no-grad --no-kuf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', '$c', round(d['ms_per_step'],3), d['breakdown_ms'], d.get('cholesky_roofline',{}).get('frac'))"
done; done | tee $O/ab.log
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
timeout 300 python tools/chol_check.py 2>&1 | tail -1 | tee $O/chol_check.log
