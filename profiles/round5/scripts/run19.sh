#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
bash tools/trace_eval.sh c4chol tools/round5/chol_once.py f32 8192
cat gpurun_out/trace_c4chol/out.txt | tail -2
grep -c . gpurun_out/trace_c4chol.txt; rm -rf gpurun_out/trace_c4chol/t
