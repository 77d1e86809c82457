# round 4, fourth GPU call: strips beside the factorisation (tests + timing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_fourth; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "beside or overlapped or one_round" > $O/pytest_overlap.log 2>&1; tail -5 $O/pytest_overlap.log
timeout 600 python tools/overlap_time.py f64 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/overlap_f64.log
timeout 600 python tools/overlap_time.py f32 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/overlap_f32.log
