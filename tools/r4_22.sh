cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_22; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.log; cat $O/time.log; cut -c1-200 $O/bench.json
for i in 1 2; do timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -3; done | tee $O/pytest_twice.log
