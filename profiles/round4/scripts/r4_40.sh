cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_40; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_grad.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
timeout 600 python tests/soak_overlap.py 2>&1 | tail -1 | tee $O/soak.log
for rep in 1 2; do
python tools/step_time.py f64 2>/dev/null | sed "s/^/new  /"
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python tools/step_time.py f64 2>/dev/null | sed "s/^/prev /"
done | tee $O/step.log
