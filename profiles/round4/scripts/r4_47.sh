cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_47; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
