#!/bin/bash
# In-profiler kernel averages of one script for several library builds:
#   tools/abprof.sh "script args" "name filter (regex)" lib1 lib2 ...
cmd=$1; flt=$2; shift 2
export TMPDIR=/tmp
for lib in "$@"; do
    if [ "$lib" = default ]; then unset LDM_HIP_LIB; else export LDM_HIP_LIB=$PWD/devlibs/lib$lib.so; fi
    d=gpurun_out/abprof/$lib; rm -rf $d; mkdir -p $d
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $cmd > $d/out.log 2>&1
    python3 - "$d" "$flt" "$lib" <<'PY'
import csv, glob, re, sys
d, flt, lib = sys.argv[1:4]
for f in glob.glob(d + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(flt, r["Name"]):
            print(f"{lib:10s} {r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
done
