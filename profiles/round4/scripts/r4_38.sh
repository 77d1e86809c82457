cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_38; mkdir -p $O
python tools/split_ab.py f64 2>/dev/null | tee $O/split_f64.log
python tools/split_ab.py f32 2>/dev/null | tee $O/split_f32.log
