cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_fifth; mkdir -p $O
for k in 1 0; do SVGP_STREAM2_LOW_PRIO=$k timeout 600 python tools/overlap_time.py f64 2>&1 | grep "n=" | sed "s/^/lowprio=$k /"; done | tee $O/overlap_prio.log
