cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
rm -rf gpurun_out/pg; SVGP_GRAD_DEBUG=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pg -- python3 tools/grad_time.py C5 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/pg/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "strip_kernel" in r["Name"]: print("debug $v", r["Calls"], round(float(r["AverageNs"])/1e6,3), "ms avg", round(float(r["MinNs"])/1e6,3), round(float(r["MaxNs"])/1e6,3))
PY
done
