#!/bin/bash
# SYRK split-K sweep (knob SVGP_GEMM_PM_SLICES = n slices on the (tiles, slices) grid): value-and-gradient time per config
mkdir -p gpurun_out/r3
for cfg in C3 C4 C5 C2; do
  for x in 0 3 7 11 15 22 30; do
    echo "slices=$x: $(SVGP_SYRK_XCD=0 SVGP_GEMM_PM_SLICES=$x python3 tools/grad_time.py $cfg | tail -1)"
  done
done 2>&1 | tee gpurun_out/r3/syrk_slices.log
