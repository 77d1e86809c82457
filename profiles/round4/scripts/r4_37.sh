cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_37; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
timeout 600 python tests/soak_overlap.py 2>&1 | tail -1 | tee $O/soak.log
for sp in 1 0; do SVGP_SEG_SPLIT=$sp python tools/mb_fwd.py f64 2>/dev/null | sed "s/^/fwd split=$sp /"; SVGP_SEG_SPLIT=$sp python tools/mb_time.py f64 2>/dev/null | sed "s/^/fwd,grad split=$sp /"; done | tee $O/ab.log
