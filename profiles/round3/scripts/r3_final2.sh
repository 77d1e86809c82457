#!/bin/bash
# round-3 closing profiles of the forward configurations (refreshes kernel_source_sha16 after the device_common.hpp additions)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for C in H C2 C4 C5; do bash tools/run_profile.sh ${C}_r3c $C > gpurun_out/prof_${C}_r3c.log 2>&1; tail -1 gpurun_out/prof_${C}_r3c.log | cut -c1-200; done
python tests/small_time.py > gpurun_out/r3/small_time.log 2>&1; tail -2 gpurun_out/r3/small_time.log
