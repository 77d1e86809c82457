cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_21; mkdir -p $O
timeout 300 python tools/chol_check.py 2>&1 | tee $O/chol_check.log
for rep in 1 2; do for c in 1 0; do SVGP_CHOL_CHAIN=$c timeout 300 python tools/prep_time.py 2>&1 | sed "s/^/chain=$c /"; done; done | tee $O/prep_ab.log
for rep in 1 2; do for c in 1 0; do for cfg in MB16k C2 C5 H; do SVGP_CHOL_CHAIN=$c timeout 300 python tools/grad_time.py $cfg 2>/dev/null | sed "s/^/chain=$c /"; done; done; done | tee $O/grad_ab.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
