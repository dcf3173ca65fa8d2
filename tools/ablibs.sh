#!/bin/bash
# Whole-benchmark and in-graph A/B of library builds inside one GPU call:  bash tools/ablibs.sh "kernel regex" lib1 lib2 ...
# ("default" = the shipped library, other names = devlibs/libNAME.so); three alternations of bench.py, then one kernel
# trace of the graph-replayed 50-step chain per build.
export TMPDIR=/tmp
re=$1; shift
for rep in 1 2 3; do for lib in "$@"; do
    if [ "$lib" = default ]; then unset LDM_HIP_LIB; else export LDM_HIP_LIB=$PWD/devlibs/lib$lib.so; fi
    timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', round(r['value'],1), 'images/s', round(r['ms_per_step'],3), 'ms')" || exit 1
done; done
for lib in "$@"; do
    if [ "$lib" = default ]; then unset LDM_HIP_LIB; else export LDM_HIP_LIB=$PWD/devlibs/lib$lib.so; fi
    d=gpurun_out/abprof/libs_$lib; rm -rf $d; mkdir -p $d
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/s64prof.py 64 > $d/out.log 2>&1 || { echo "$lib trace failed"; exit 1; }
    echo "== $lib"; python3 tools/kstats.py $d/*/*kernel_stats.csv 14 | grep -E "$re"
done
