"""Per (kernel, grid size) statistics of a rocprofv3 kernel_trace.csv, in order of first appearance.  With a chunk size
the launches of each group are cut into consecutive chunks (one per benchmark case when every case launches the kernel
the same number of times) and each chunk's median is printed.
  python3 tools/ktrace_groups.py path/to/kernel_trace.csv [name filter] [chunk]"""
import collections, csv, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
acc = collections.OrderedDict()
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    if flt not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"].replace("void ldm::", "").replace("ldm::", "")[:60], r["Grid_Size_X"], r.get("Grid_Size_Y", ""))
    acc.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, gx, gy), v in acc.items():
    s = sorted(v)
    line = f"{n:60s} grid {gx:>8s} x {gy:>5s}  calls {len(v):4d}  median {s[len(s)//2]/1e3:8.2f} us  min {s[0]/1e3:8.2f}"
    if chunk:
        meds = [sorted(v[i:i + chunk])[len(v[i:i + chunk]) // 2] / 1e3 for i in range(0, len(v), chunk)]
        line += "  chunks " + " ".join(f"{m:.1f}" for m in meds)
    print(line)
