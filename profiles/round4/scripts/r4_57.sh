cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_57; mkdir -p $O
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_cnt64.so timeout 300 python tools/chol_check.py 2>&1 | tail -1 | tee $O/chol_check.log
for rep in 1 2 3; do
python tools/prep_time.py 2>/dev/null | grep f64 | sed "s/^/cnt32 /"
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_cnt64.so python tools/prep_time.py 2>/dev/null | grep f64 | sed "s/^/cnt64 /"
done | tee $O/prep_ab.log
