#!/usr/bin/env bash
# timing-only ablation builds of the strip kernels (tools/build_ablate.sh strip N; WRONG results) under rocprofv3: the value-and-gradient strip
# launch's average duration per build.  bits: 32 no point-major A, 64 no point-major R A, 128 no K-dot
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in "$@"; do
  if [ "$V" = 0 ]; then L=$GRAFT_REPO_ROOT/approximategps.jl_amd/csrc/libsvgp_mi355x.so; else L=$GRAFT_REPO_ROOT/approximategps.jl_amd/csrc/ablate/libsvgp_strip_$V.so; fi
  O=/tmp/abl_$V; rm -rf $O; mkdir -p $O
  SVGP_MI355X_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/grad_time.py ${CFG:-H} > $O/out.txt 2> $O/err.txt
  f=$(ls -S $O/stats/*/*kernel_stats.csv | head -1)
  echo "ablate=$V $(grep elbo_grad $O/out.txt | cut -c1-40) $(python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "strip_kernel" in n and ", true," in n.split("strip_kernel")[1][:40] and int(r["Calls"]) >= 8:
        print(f'grad strips avg {float(r["AverageNs"])/1e3:.1f} us', end=" ")
    if "syrk_async" in n: print(f'syrk {float(r["AverageNs"])/1e3:.1f} us', end=" ")
PY
)"
done
