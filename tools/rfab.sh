#!/bin/bash
# Fused ResidualBlock kernel against the two launches it replaces: parity test, in-graph kernel averages, benchmark A/B.
export TMPDIR=/tmp
timeout -k 10 120 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 60 --timeout-method thread -k "resblock_fused" 2>&1 | tail -2 || exit 1
for v in 1 0; do
    d=gpurun_out/abprof/rf$v; rm -rf $d; mkdir -p $d
    LDM_RESFUSE=$v timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/s64prof.py 64 > $d/out.log 2>&1
    echo RESFUSE=$v; python3 tools/kstats.py $d/*/*kernel_stats.csv 12 | grep "resfuse\|ws64_kernel<[12]>"
done
for rep in 1 2 3; do for v in 1 0; do
    LDM_RESFUSE=$v timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('RESFUSE=$v', round(r['value'],1), round(r['ms_per_step'],3))"
done; done
