#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
bash tools/run_profile.sh H_r3b H > gpurun_out/prof_H_r3b.log 2>&1
bash tools/run_profile.sh Hgrad_r3b H grad > gpurun_out/prof_Hgrad_r3b.log 2>&1
bash tools/run_profile.sh C5grad_r3b C5 grad > gpurun_out/prof_C5grad_r3b.log 2>&1
bash tools/run_profile.sh C4_r3b C4 > gpurun_out/prof_C4_r3b.log 2>&1
STEPS=10 bash tools/run_all.sh 2>/dev/null | tail -7
python tests/small_time.py > gpurun_out/r3/small_time.log 2>&1; tail -1 gpurun_out/r3/small_time.log
bash tools/trace_eval.sh mb16k_final tools/mb_grad.py 16384 1024 8
