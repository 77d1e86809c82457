#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3/pt.log 2>&1; grep -E "passed|failed|error" gpurun_out/r3/pt.log | tail -3
for cfg in C2 H C5; do python3 tools/grad_time.py $cfg | tail -1; done
bash tools/trace_eval.sh mb16k_r3d tools/mb_grad.py 16384 1024 8
