cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_22; mkdir -p $O
for rep in 1 2; do for c in 1 0; do SVGP_CHOL_CHAIN=$c timeout 300 python tools/prep_time.py 2>&1 | sed "s/^/chain=$c /"; done; done | tee $O/prep_ab.log
cd /tmp; SVGP_CHOL_CHAIN=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_chain -o chain -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py MB16k > /dev/null 2>&1
SVGP_CHOL_CHAIN=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_nochain -o nochain -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py MB16k > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for f in $O/prof_chain/*kernel_stats.csv $O/prof_nochain/*kernel_stats.csv; do echo $f; grep -i "chol\|potf2" $f | cut -c1-60,100-400; done | tee $O/stats.log
rm -f $O/prof_*/*kernel_trace.csv $O/prof_*/*agent* 
