cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_15; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "beside or overlapped or rounds" > $O/pytest.log 2>&1; grep -n "passed\|failed\|^E " $O/pytest.log | tail -8
for rep in 1 2; do for c in C5 C2 H32 C3; do for k in 1 0; do SVGP_OVERLAP_HEAD=$k python tools/ablate_time.py $c 2>/dev/null | sed "s/^/head=$k /"; done; done; done | tee $O/head_ab.log
for k in 1 0; do SVGP_OVERLAP_HEAD=$k python tools/ab_r3.py 2>&1 | grep "n=" | sed "s/^/head=$k /"; done | tee $O/head_wall.log
