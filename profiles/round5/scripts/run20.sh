#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_syrkocc; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
timeout 600 python tools/round5/chol_once.py f32 8192 2>&1 | tail -1
timeout 600 python tools/round5/chol_once.py f32 4096 2>&1 | tail -1
timeout 600 python tools/round5/chol_once.py f32 2304 2>&1 | tail -1
done
bash tools/trace_eval.sh c4chol2 tools/round5/chol_once.py f32 8192; rm -rf gpurun_out/trace_c4chol2/t
grep "syrk128" gpurun_out/trace_c4chol2.txt | awk '{print $3}' | tr '\n' ' '
