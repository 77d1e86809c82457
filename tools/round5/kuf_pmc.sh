#!/usr/bin/env bash
# SQ counters of the standalone Kuf kernels (tools/kuf_time.py), averaged per kernel name
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_kufpmc; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- python3 tools/kuf_time.py "$@" > $O/out.log 2> $O/err.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/sq2 -- python3 tools/kuf_time.py "$@" > $O/out2.log 2> $O/err2.log
python3 - <<'PY'
import csv, glob, collections
for sub in ("sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r5_kufpmc/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "kuf" not in k: continue
            k = k[k.index("kuf"):][:40]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(acc.items()):
        print(k, {n: f"{sum(v)/len(v):.4g}" for n, v in sorted(c.items())}, "launches", len(next(iter(c.values()))))
PY
find $O -name "*.csv" -size +2M -delete
