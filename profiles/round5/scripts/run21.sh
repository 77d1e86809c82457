#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kgrad; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
timeout 900 python -m pytest tests/test_gpu_round4.py -k wide_inputs -m gpu -q -x 2>&1 | f | tail -n 3
timeout 900 python tools/grad_time.py Hd32 Hd64 H32d32 H32d64 2>&1 | f | tee $O/grad_time_${1:-a}.log | cut -c1-200
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_${1:-a} -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py Hd64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob
for f in glob.glob("$O/stats_${1:-a}/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
rm -rf $O/stats_${1:-a}
