#!/bin/bash
# Whole-benchmark A/B of library builds on another benchmark configuration:
#   bash tools/ablibs_cfg.sh "bench.py flags" lib1 lib2 ...      ("default" = the shipped library)
flags=$1; shift
for rep in 1 2 3; do for lib in "$@"; do
    if [ "$lib" = default ]; then unset LDM_HIP_LIB; else export LDM_HIP_LIB=$PWD/devlibs/lib$lib.so; fi
    timeout -k 10 400 python3 bench.py $flags --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', round(r['value'],2), r['unit'], round(r['ms_per_step'],2), 'ms')" || exit 1
done; done
