cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_36; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/pytest.log
for rep in 1 2; do for sp in 1 0; do for dt in f64 f32; do SVGP_SEG_SPLIT=$sp python tools/mb_fwd.py $dt 2>/dev/null | sed "s/^/fwd split=$sp /"; done; done; done | tee $O/fwd.log
for rep in 1 2; do for sp in 1 0; do SVGP_SEG_SPLIT=$sp python tools/mb_time.py f32 2>/dev/null | sed "s/^/f32 fwd,grad split=$sp /"; done; done | tee $O/f32.log
