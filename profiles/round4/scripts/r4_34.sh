cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_34; mkdir -p $O
python tools/mb_one.py 1024 1024 2>/dev/null | tee $O/times.log
python tools/mb_one.py 2048 512 2>/dev/null | tee -a $O/times.log
cd /tmp
for sh in "1024 1024" "2048 512"; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/mb_one.py $sh > /dev/null 2>&1
f=$(ls $O/prof/*kernel_trace.csv | head -1)
echo "== $sh"; python3 $GRAFT_REPO_ROOT/tools/trace_eval.py $f
rm -rf $O/prof
done > $O/trace.log 2>&1
