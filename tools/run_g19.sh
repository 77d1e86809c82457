cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py -m gpu -q 2>&1 | tail -1
for rep in 1 2; do for cfg in H H32 C3; do
  SVGP_PREFETCH=1 python tools/ablate_time.py $cfg 2>/dev/null | sed "s/^/pf1 /"
  SVGP_PREFETCH=0 python tools/ablate_time.py $cfg 2>/dev/null | sed "s/^/pf0 /"
done; done
