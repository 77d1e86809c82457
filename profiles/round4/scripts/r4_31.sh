cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_31; mkdir -p $O
for rep in 1 2; do
python tools/step_time.py f64 2>&1 | tail -5 | sed "s/^/new reuse  /"
STEP_REUSE_OUT=0 python tools/step_time.py f64 2>&1 | tail -5 | sed "s/^/new fresh  /"
done | tee $O/step.log
