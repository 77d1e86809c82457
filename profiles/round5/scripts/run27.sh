#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
bash tools/trace_eval.sh mb16kgrad tools/mb_grad.py 16384 1024 8
tail -2 gpurun_out/trace_mb16kgrad/out.txt; rm -rf gpurun_out/trace_mb16kgrad/t
bash tools/trace_eval.sh c5grad tools/mb_grad.py 262144 1024 8 f32
tail -2 gpurun_out/trace_c5grad/out.txt; rm -rf gpurun_out/trace_c5grad/t
