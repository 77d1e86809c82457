"""All launches of the LAST evaluation in a rocprofv3 kernel trace (us relative to its first block factorisation)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "potf2_kernel" in r["Kernel_Name"]]
i0 = idx[-1]
while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 30000: i0 -= 1   # back to the first launch of the call
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    n = r["Kernel_Name"].replace("svgp::(anonymous namespace)::", "").replace("void ", "")[:56]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{n:56s} wg {int(r["Grid_Size_X"])//max(int(r["Workgroup_Size_X"]),1):5d}  start {(st-t0)/1e3:8.1f}  dur {(en-st)/1e3:7.1f}  gap {(st-prev_end)/1e3:6.1f}')
    prev_end = max(prev_end, en)
