#!/bin/bash
# round-3 closing run: GPU suite, gradient profiles (stats + PMC passes), every config, minibatch trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3) > gpurun_out/r3/pytest_final.log 2>&1
cat gpurun_out/r3/pytest_final.log
bash tools/run_profile.sh Hgrad_r3c H grad > gpurun_out/prof_Hgrad_r3c.log 2>&1
bash tools/run_profile.sh C5grad_r3c C5 grad > gpurun_out/prof_C5grad_r3c.log 2>&1
STEPS=10 bash tools/run_all.sh 2>/dev/null | tail -7
bash tools/trace_eval.sh mb16k_r3c tools/mb_grad.py 16384 1024 8
