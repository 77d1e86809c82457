cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ninth; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -3
for rep in 1 2; do for c in H H32 C5 C2; do for u in 1 0; do SVGP_SYRK_UNIFORM=$u python tools/grad_time.py $c 2>/dev/null | grep elbo_grad | sed "s/^/uniform=$u /"; done; done; done | tee $O/grad_time.log
