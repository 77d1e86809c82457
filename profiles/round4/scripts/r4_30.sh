cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_30; mkdir -p $O
ls -la approximategps.jl_amd/csrc/ablate/ | head
for rep in 1 2; do
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_prev.so python tools/step_time.py f64 2>&1 | tail -5 | sed "s/^/prev /"
python tools/step_time.py f64 2>/dev/null | sed "s/^/new  /"
done | tee $O/step.log
