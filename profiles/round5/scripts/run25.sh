#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_kgrad; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
for rep in 1 2; do
for lib in default kgkv1; do
  if [ $lib = default ]; then unset SVGP_MI355X_LIB; else export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_$lib.so; fi
  echo "== $lib"
  for c in C3 Hd16; do timeout 900 python tools/grad_time.py $c 2>&1 | f | grep elbo_grad | tee -a $O/kgrad_kv1_ab.log | cut -c1-200; done
done; done
export SVGP_MI355X_LIB=$PWD/approximategps.jl_amd/csrc/ablate/libsvgp_kgkv1.so
timeout 900 python -m pytest tests/test_gpu_grad.py -m gpu -q -x 2>&1 | f | tail -n 2
unset SVGP_MI355X_LIB
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c3 -- python3 $GRAFT_REPO_ROOT/tools/grad_time.py C3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob
for f in glob.glob("$O/stats_c3/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
rm -rf $O/stats_c3
