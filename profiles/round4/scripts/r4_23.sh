cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_23; mkdir -p $O
cd /tmp
for c in 1 0; do
export SVGP_CHOL_CHAIN=$c
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$c -o t -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py MB16k > /dev/null 2>&1
f=$(ls $O/prof_$c/*kernel_trace.csv | head -1)
echo "chain=$c"; python3 $GRAFT_REPO_ROOT/tools/trace_chain.py $f
done | tee $O/trace.log
rm -rf $O/prof_*
