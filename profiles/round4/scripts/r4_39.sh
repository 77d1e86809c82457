cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_39; mkdir -p $O
python tools/minpanels_ab.py f64 2>/dev/null | tee $O/minpanels_f64.log
python tools/minpanels_ab.py f32 2>/dev/null | tee $O/minpanels_f32.log
