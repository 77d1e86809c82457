cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py -m gpu -q -x 2>&1 | tail -3)
for rep in 1 2; do for cfg in H32 C3 C4 C5; do
  SVGP_F32_NT=128 python tools/ablate_time.py $cfg 2>/dev/null | sed "s/^/f32nt128 /"
  SVGP_F32_NT=64 python tools/ablate_time.py $cfg 2>/dev/null | sed "s/^/f32nt64  /"
done; done
