cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_51; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
timeout 900 python tests/soak_overlap.py 300 2>&1 | tail -1 | tee $O/soak.log
timeout 600 python tests/soak.py 2>&1 | tail -2 | tee $O/soak2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; cut -c1-200 $O/bench_default.json
