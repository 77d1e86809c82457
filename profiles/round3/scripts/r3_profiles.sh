#!/bin/bash
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -5) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for C in H C2 C4 C5; do bash tools/run_profile.sh ${C}_r3 $C > gpurun_out/prof_${C}_r3.log 2>&1; tail -1 gpurun_out/prof_${C}_r3.log | cut -c1-200; done
bash tools/run_profile.sh Hgrad_r3 H grad > gpurun_out/prof_Hgrad_r3.log 2>&1
bash tools/run_profile.sh C5grad_r3 C5 grad > gpurun_out/prof_C5grad_r3.log 2>&1
python tests/small_time.py > gpurun_out/r3/small_time.log 2>&1; tail -2 gpurun_out/r3/small_time.log
