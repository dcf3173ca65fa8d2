"""Print a rocprofv3 kernel_stats.csv as percentages (argument: path)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    n = r["Name"].replace("ldm::", "").replace("void ", "")[:100]
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls={r['Calls']:>5} avg={float(r['AverageNs'])/1e3:7.1f}us  {n}")
