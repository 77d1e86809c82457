cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_55; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_grad.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
for rep in 1 2 3; do
python tools/mb_fwd.py f64 2>/dev/null | sed "s/^/new  fwd  /"
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python tools/mb_fwd.py f64 2>/dev/null | sed "s/^/prev fwd  /"
done | tee $O/ab.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py MB16k > /dev/null 2>&1
f=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_eval.py $f | head -8 > $O/trace_head.log 2>&1
rm -rf $O/prof
