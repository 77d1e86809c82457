cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_29; mkdir -p $O
for rep in 1 2; do
python tools/step_time.py f64 2>/dev/null | sed "s/^/new  /"
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python tools/step_time.py f64 2>/dev/null | sed "s/^/prev /"
done | tee $O/step.log
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_round4.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.log
