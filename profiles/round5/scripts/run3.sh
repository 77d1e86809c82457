#!/usr/bin/env bash
# round 5, third GPU visit: the GPU suite on the product library and on the experiments library; fp32 value gap; bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu -x > gpurun_out/r5/gputest_product.log 2>&1; tail -n 3 gpurun_out/r5/gputest_product.log
SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so python -m pytest tests -q -m gpu > gpurun_out/r5/gputest_experiments.log 2>&1; tail -n 3 gpurun_out/r5/gputest_experiments.log
python tests/f32_value_gap.py > gpurun_out/r5/f32_value_gap.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5/f32_value_gap.log
python bench.py > gpurun_out/r5/bench3.json 2> gpurun_out/r5/bench3.err; tail -c 600 gpurun_out/r5/bench3.json
