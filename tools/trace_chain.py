"""Start/end of the factorisation launches of the LAST prep in a rocprofv3 kernel trace (us relative to the first of them)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "potf2_kernel" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 24]:
    n = r["Kernel_Name"].replace("svgp::(anonymous namespace)::", "").replace("void ", "")[:48]
    print(f'{n:48s} grid {int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0))//256:4d}  start {(int(r["Start_Timestamp"])-t0)/1e3:8.1f}  end {(int(r["End_Timestamp"])-t0)/1e3:8.1f}  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:6.1f}')
