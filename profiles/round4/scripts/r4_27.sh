cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_27; mkdir -p $O
for rep in 1 2; do for r in 0 32 48 64 96 128; do for dt in f64 f32; do SVGP_STREAM2_RESERVE=$r timeout 300 python tools/mb_time.py $dt 2>/dev/null | sed "s/^/reserve=$r /"; done; done; done | tee $O/reserve.log
