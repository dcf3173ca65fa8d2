"""Median GPU-side duration per (kernel, grid) from a rocprofv3 kernel_trace.csv."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:90], r.get("Grid_Size_X") or r.get("Grid_Size"))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k[0][-70:]} grid={k[1]:>8} n={len(v)} median={v[len(v)//2]:.1f}us min={v[0]:.1f}")
