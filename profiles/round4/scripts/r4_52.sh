cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_52; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
for rep in 1 2 3; do for f in 1 0; do
SVGP_ROW_EVENT_EXT=$f python tools/mb_fwd.py f64 2>/dev/null | sed "s/^/ext=$f fwd  /"
SVGP_ROW_EVENT_EXT=$f python tools/mb_time.py f64 2>/dev/null | sed "s/^/ext=$f f,g  /"
done; done | tee $O/ext.log
timeout 900 python tests/soak_overlap.py 300 2>&1 | tail -1 | tee $O/soak.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py MB16k > /dev/null 2>&1
f=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_eval.py $f > $O/trace_MB16k_grad.log 2>&1
rm -rf $O/prof
