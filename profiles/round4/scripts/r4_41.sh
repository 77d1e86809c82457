cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_41; mkdir -p $O
for rep in 1 2 3; do
python tools/step_time.py f64 2>/dev/null | sed "s/^/splitk=1 /"
SVGP_GEMM_MM_SPLITK=0 python tools/step_time.py f64 2>/dev/null | sed "s/^/splitk=0 /"
done | tee $O/step.log
