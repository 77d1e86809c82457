cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "gradient_strips_beside" > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest.log | tail -5
timeout 600 python tools/overlap_grad_time.py f64 2>&1 | grep "grad n=" | tee $O/overlap_grad_f64.log
timeout 600 python tools/overlap_grad_time.py f32 2>&1 | grep "grad n=" | tee $O/overlap_grad_f32.log
