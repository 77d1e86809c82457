cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_25; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_grad.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
for rep in 1 2; do
for cfg in MB4k MB16k C2 C5; do
  SVGP_OVERLAP=0 timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/ov=0        /"
  timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/ov=1        /"
  SVGP_OVERLAP_HEAD=1 timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/ov=1 head=1 /"
done; done | tee $O/ab.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py MB16k > /dev/null 2>&1
f=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_eval.py $f > $O/trace_MB16k_grad.log 2>&1
rm -rf $O/prof
