#!/bin/bash
# In-graph kernel averages of one 50-step call at [64,64,64,64] under rocprofv3, once per value of an environment switch:
#   tools/kprof_env.sh VAR "v1 v2 ..." "name filter (regex)"
var=$1; vals=$2; flt=${3:-.}
export TMPDIR=/tmp
for v in $vals; do
    d=gpurun_out/kprof_env/$var=$v; rm -rf "$d"; mkdir -p "$d"
    export "$var=$v"
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/s64prof.py 64 bf16 > "$d/out.log" 2>&1
    python3 - "$d" "$flt" "$var=$v" <<'PY'
import csv, glob, re, sys
d, flt, tag = sys.argv[1:4]
tot = 0.0
for f in glob.glob(d + "/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        tot += float(r["TotalDurationNs"])
        if re.search(flt, r["Name"]):
            print(f"{tag:18s} {r['Name'][:84]:84s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
print(f"{tag:18s} total kernel time {tot/1e6:.2f} ms")
PY
done
