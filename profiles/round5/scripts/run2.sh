#!/usr/bin/env bash
# round 5, second GPU visit: trailing-kgrad pipeline mode; potf2 with SGPR broadcasts for the bulk terms (stamps builds)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
python tools/round5/pipe_ab.py H,H32,C5,C3 5 "1:1:0:1,2:1:0:2,2:1:1:2,2:1:-1:2,3:1:0:2" > gpurun_out/r5/pipe_ab2.log 2>&1
for lib in stamps sgpr_stamps; do
  for rep in 1 2; do
    SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_$lib.so python tools/potf2_time.py >> gpurun_out/r5/potf2_sgpr.log 2>&1
  done
done
cat gpurun_out/r5/pipe_ab2.log; cat gpurun_out/r5/potf2_sgpr.log
python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r5/dist_a.log 2>&1; tail -3 gpurun_out/r5/dist_a.log
SVGP_GRAD_PIPELINE=0 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r5/dist_b.log 2>&1; tail -3 gpurun_out/r5/dist_b.log
python -m pytest tests -q -m gpu --deselect "tests/test_gpu_distributed.py::test_library_collective_world_of_one" > gpurun_out/r5/gputest2.log 2>&1; tail -5 gpurun_out/r5/gputest2.log
