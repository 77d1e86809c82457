#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kgrad; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_experiments.so
for v in 0 1 0 1; do
  echo "== SVGP_KGRAD_WIDE2=$v"
  for c in Hd32 Hd64 H32d32 H32d64; do SVGP_KGRAD_WIDE2=$v timeout 900 python tools/grad_time.py $c 2>&1 | f | grep elbo_grad | tee -a $O/kgrad_wide2_ab.log | cut -c1-200; done
done
for v in 0 1; do
cd /tmp; SVGP_KGRAD_WIDE2=$v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$v -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py Hd64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob
for f in glob.glob("$O/stats_$v/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print("$v", r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
rm -rf $O/stats_$v
done
