#!/bin/bash
# same-box A/B: f64 strips as 2 x (256 threads, 128 x 64 tile) vs 1 x (512 threads, 128 x 128 tile), both on the
# asynchronous three-buffer LDS-DMA loop; then PMC counters for both
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
B="bench.py --config H --no-cpu-baseline --no-grad --no-c5 --no-kuf"
for i in 1 2; do
  for nt in 64 128; do
    echo "SVGP_STRIP_NT=$nt"; SVGP_STRIP_NT=$nt python $B --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['breakdown_ms']['strip'],3), 'ms strip', round(d['roofline']['frac'],4), d['config']['elbo'])"
  done
done 2>&1 | tee gpurun_out/r3/strip_ab.log
for nt in 64 128; do
  O=gpurun_out/r3/strip_pmc_$nt; rm -rf $O; mkdir -p $O
  SVGP_STRIP_NT=$nt rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -- python3 $B --steps 3 --warmup 1 > /dev/null 2> $O/sq.err
  SVGP_STRIP_NT=$nt rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $B --steps 3 --warmup 1 > /dev/null 2> $O/fetch.err
  python3 - $O $nt <<'PY' | tee -a gpurun_out/r3/strip_ab.log
import csv, glob, sys
from collections import defaultdict
O, nt = sys.argv[1], sys.argv[2]
out = {}
for sub in ("sq", "fetch"):
    acc = defaultdict(list)
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        disp = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "strip_kernel" in r["Kernel_Name"]:
                disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        for c in disp.values():
            for k, v in c.items():
                acc[k].append(v)
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
if "GRBM_GUI_ACTIVE" in out:
    out["mfma_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / (out["GRBM_GUI_ACTIVE"] * 1024 / 8)
    out["wait_frac"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in out:
    out["fetch_GB_corrected"] = 2 * 1024 * out["FETCH_SIZE"] / 1e9
print("NT", nt, {k: (round(v, 4) if v < 1e4 else f"{v:.4g}") for k, v in out.items()})
PY
  rm -rf $O
done
