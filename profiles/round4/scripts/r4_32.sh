cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_32; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
timeout 600 python tests/soak_overlap.py 2>&1 | tail -3 | tee $O/soak.log
