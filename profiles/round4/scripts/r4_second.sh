# round 4, second GPU call: the point-gradient kernel (SVGP_GRAD_POST 1 / 0) and the wide-input pre-generation, tests first
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_second; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for rep in 1 2; do for c in H H32 C2 C5; do for k in 1 0; do
  SVGP_GRAD_POST=$k python tools/grad_time.py $c 2>/dev/null | sed "s/^/post=$k /"
done; done; done > $O/grad_post_ab.log; cat $O/grad_post_ab.log
for c in Hd17 Hd32 Hd64 H32d32 H32d64; do for k in 1 0; do
  SVGP_PREGEN_MFMA_BIGD=$k python tools/ablate_time.py $c 2>/dev/null | sed "s/^/bigd=$k /"
done; done > $O/wide_d_ab.log; cat $O/wide_d_ab.log
python tools/grad_time.py C3 2>/dev/null | tee $O/grad_C3.log
python tools/grad_time.py Hd32 2>/dev/null | tee $O/grad_Hd32.log
