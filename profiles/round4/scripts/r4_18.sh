cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_18; mkdir -p $O
SVGP_OVERLAP_DRY=1 timeout 300 python tools/overlap_time.py f64 > $O/dry1.log 2>&1; tail -12 $O/dry1.log
