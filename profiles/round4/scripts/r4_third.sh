# round 4, third GPU call: full GPU tests with the fixed fp32 gemv, then the value-and-gradient profiles (H, C5)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_third; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/run_profile.sh r4_Hgrad H grad > $O/prof_H.log 2>&1
bash tools/run_profile.sh r4_C5grad C5 grad > $O/prof_C5.log 2>&1
