cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_56; mkdir -p $O
timeout 300 python tools/chol_check.py 2>&1 | tail -11 | tee $O/chol_check.log
for rep in 1 2 3; do
python tools/prep_time.py 2>/dev/null | sed "s/^/new  /"
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python tools/prep_time.py 2>/dev/null | sed "s/^/prev /"
done | tee $O/prep_ab.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py tests/test_gpu_round4.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
