cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_24; mkdir -p $O
cd /tmp
for cfg in MB16k C2; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$cfg -o t -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py $cfg > /dev/null 2>&1
f=$(ls $O/prof_$cfg/*kernel_trace.csv | head -1)
echo "== $cfg"; python3 $GRAFT_REPO_ROOT/tools/trace_eval.py $f
done > $O/trace.log 2>&1
rm -rf $O/prof_*
