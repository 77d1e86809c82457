#!/bin/bash
# async SYRK A/B (knob SVGP_SYRK_ASYNC), value-and-gradient time per config
mkdir -p gpurun_out/r3
timeout 900 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for cfg in H C2 C5 C3 C4 H32; do
  for a in 0 1; do
    echo "async=$a: $(SVGP_SYRK_ASYNC=$a python3 tools/grad_time.py $cfg | tail -1)"
  done
done 2>&1 | tee gpurun_out/r3/syrk_async.log
