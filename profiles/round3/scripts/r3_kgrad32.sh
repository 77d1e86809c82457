#!/bin/bash
# fp32 kgrad with fp32 block sums (build knob SVGP_KGRAD_F32_BLOCKS) against fp64 accumulation of every entry: tests, accuracy, time
mkdir -p gpurun_out/r3
A=approximategps.jl_amd/csrc/ablate
timeout 900 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3/pt.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r3/pt.log | tail -3
for cfg in C5 H32 C3 C4; do
  echo "blocks: $(python3 tools/grad_time.py $cfg | tail -1)"
  echo "f64acc: $(SVGP_MI355X_LIB=$A/libsvgp_f64acc.so python3 tools/grad_time.py $cfg | tail -1)"
done 2>&1 | tee gpurun_out/r3/kgrad_f32_blocks.log
timeout 600 python3 tests/f32_grad_accuracy.py > gpurun_out/r3/f32acc_blocks.log 2>&1; cat gpurun_out/r3/f32acc_blocks.log | tail -30
