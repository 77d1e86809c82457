cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_48; mkdir -p $O
POTF2_DTYPES=f64 SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_stamps.so python tools/potf2_time.py 2>/dev/null | tee $O/stamps_f64.log
POTF2_DTYPES=f32 SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_stamps.so python tools/potf2_time.py 2>/dev/null | tee $O/stamps_f32.log
