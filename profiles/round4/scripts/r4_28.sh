cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_28; mkdir -p $O
python tools/mb_one.py 32768 1024 2>/dev/null | tee $O/times.log
python tools/mb_one.py 4096 2048 2>/dev/null | tee -a $O/times.log
SVGP_OVERLAP=0 python tools/mb_one.py 32768 1024 2>/dev/null | sed "s/^/ov=0 /" | tee -a $O/times.log
SVGP_OVERLAP=0 python tools/mb_one.py 4096 2048 2>/dev/null | sed "s/^/ov=0 /" | tee -a $O/times.log
cd /tmp
for sh in "32768 1024" "4096 2048"; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/mb_one.py $sh > /dev/null 2>&1
f=$(ls $O/prof/*kernel_trace.csv | head -1)
echo "== $sh"; python3 $GRAFT_REPO_ROOT/tools/trace_eval.py $f
rm -rf $O/prof
done > $O/trace.log 2>&1
