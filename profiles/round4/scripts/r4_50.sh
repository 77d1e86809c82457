cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_50; mkdir -p $O
for rep in 1 2; do
python tools/mb_time.py f64 2>/dev/null | sed "s/^/new  f,g  /"
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python tools/mb_time.py f64 2>/dev/null | sed "s/^/prev f,g  /"
done | tee $O/ab.log
python tools/split_ab.py f64 2>/dev/null | tee $O/split_f64.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py MB16k > /dev/null 2>&1
f=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_eval.py $f > $O/trace_MB16k_grad.log 2>&1
rm -rf $O/prof
