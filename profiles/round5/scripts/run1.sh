#!/usr/bin/env bash
# first GPU visit of round 5: the GPU suite on the pipelined library, the pipeline A/B, a bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
python tools/round5/pipe_ab.py H,H32,C5,C2,C3 5 > gpurun_out/r5/pipe_ab.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest1.log 2>&1
python bench.py > gpurun_out/r5/bench1.json 2> gpurun_out/r5/bench1.err
tail -3 gpurun_out/r5/gputest1.log; cat gpurun_out/r5/pipe_ab.log
