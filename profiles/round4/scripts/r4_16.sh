cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_16; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "beside or overlapped or rounds" > $O/pytest.log 2>&1; grep -n "passed\|failed\|^E " $O/pytest.log | tail -8
for k in 1 0 1 0; do SVGP_OVERLAP_P2CKPT=$k timeout 600 python tools/overlap_time.py f64 2>&1 | grep "n=" | sed "s/^/ckpt=$k /"; done | tee $O/ckpt_f64.log
for k in 1 0; do SVGP_OVERLAP_P2CKPT=$k timeout 600 python tools/overlap_time.py f32 2>&1 | grep "n=" | sed "s/^/ckpt=$k /"; done | tee $O/ckpt_f32.log
