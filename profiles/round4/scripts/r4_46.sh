cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_46; mkdir -p $O
for rep in 1 2 3; do for t in 1 0; do
SVGP_TIMING=$t python tools/mb_fwd.py f64 2>/dev/null | sed "s/^/timing=$t fwd  /"
SVGP_TIMING=$t python tools/mb_time.py f64 2>/dev/null | sed "s/^/timing=$t f,g  /"
done; done | tee $O/timing.log
