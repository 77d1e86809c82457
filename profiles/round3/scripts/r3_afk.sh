#!/bin/bash
# kgrad loads in flight (build knob SVGP_KGRAD_U) A/B + the gradient tests on the default build
mkdir -p gpurun_out/r3
timeout 900 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_distributed.py -x -q > gpurun_out/r3/pt.log 2>&1; tail -3 gpurun_out/r3/pt.log
A=approximategps.jl_amd/csrc/ablate
for cfg in H C5 C2; do
  for lib in "" $A/libsvgp_u8.so $A/libsvgp_u2.so; do
    echo "lib=$lib: $(SVGP_MI355X_LIB=$lib python3 tools/grad_time.py $cfg | tail -1)"
  done
done 2>&1 | tee gpurun_out/r3/kgrad_u.log
SVGP_MI355X_LIB=$A/libsvgp_u8.so bash tools/trace_eval.sh hgrad8 tools/grad_time.py H
