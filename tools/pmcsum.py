"""Print per-kernel averages of a rocprofv3 --pmc counter_collection.csv (argument: path [name filter])."""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (kn, cn), (v, n) in sorted(acc.items()):
    if flt in kn:
        print(f"{kn:70s} {cn:32s} launches={n:4d} avg={v/n:16.1f}")
