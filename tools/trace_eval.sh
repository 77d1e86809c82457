#!/bin/bash
# kernel timeline of ONE value-and-gradient (or forward) evaluation: tools/trace_eval.sh <tag> <python script + args...>
# writes gpurun_out/trace_<tag>.txt: per launch  start offset (us), duration (us), gap to the previous kernel's end (us), name
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/trace_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 "$@" > $O/out.txt 2> $O/err.txt
f=$(ls -S $O/t/*/*kernel_trace.csv | head -1)
python3 - "$f" gpurun_out/trace_$TAG.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last evaluation = everything from the last scale_inputs_kernel on
idx = [i for i, r in enumerate(rows) if "scale_inputs_kernel" in r["Kernel_Name"]]
rows = rows[idx[-1]:] if idx else rows
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
with open(sys.argv[2], "w") as f:
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = re.sub(r"void svgp::\(anonymous namespace\)::|svgp::\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n)[:70]
        f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {n}  grid={r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg={r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}\n")
        prev_end = max(prev_end, e)
    f.write(f"total {(prev_end - t0) / 1e3:.1f} us, {len(rows)} launches\n")
PY
rm -rf $O/t
tail -1 gpurun_out/trace_$TAG.txt
