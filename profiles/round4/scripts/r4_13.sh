cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_13; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; grep -n "^FAILED\|passed\|failed" $O/pytest_full.log | tail -8
