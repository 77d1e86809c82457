# PMC passes over the Kuf kernel alone (tools/kuf_time.py): SQ issue/wait split, store-path stalls
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kufpmc_$1; C=${2:-H}; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq1 -- python3 tools/kuf_time.py $C > $O/a.txt 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/sq2 -- python3 tools/kuf_time.py $C > $O/b.txt 2> $O/b.err
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/tcc -- python3 tools/kuf_time.py $C > $O/c.txt 2> $O/c.err
python3 - <<PY
import csv, glob, collections
for sub in ("sq1","sq2","tcc"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "kuf" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(sub, k, {n: round(sum(v)/len(v)) for n, v in c.items()})
PY
tail -3 $O/c.err | cut -c1-200
