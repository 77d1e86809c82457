cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_53; mkdir -p $O
timeout 2400 python tests/soak_overlap.py 2400 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tee $O/soak_overlap_long.log
