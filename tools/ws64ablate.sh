#!/bin/bash
# Timing ablations of the 64-channel weight-stationary kernel inside the graph-replayed 50-step DDIM (B = 64), all four
# epilogue modes (LDM_RESFUSE=0 keeps the ResidualBlocks on it): the shipped library against development builds with
# parts compiled out (tools/devbuild.sh abN "-DWS64_ABLATE=N": 1 no window DMA after the first tile, 2 no output
# stores, 4 no staging of the epilogue's inputs, 7 all three = MFMA + LDS reads only).  Ablated builds compute wrong
# results; only their kernel durations mean anything.   bash tools/ws64ablate.sh > gpurun_out/ws64_ablation.txt
export TMPDIR=/tmp
for lib in default ab1 ab2 ab4 ab7; do
    if [ "$lib" = default ]; then unset LDM_HIP_LIB; else export LDM_HIP_LIB=$PWD/devlibs/lib$lib.so; fi
    d=gpurun_out/abprof/ws64_$lib; rm -rf $d; mkdir -p $d
    LDM_RESFUSE=0 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/s64prof.py 64 > $d/out.log 2>&1 || { echo "$lib failed"; tail -3 $d/out.log; exit 1; }
    echo "== $lib"
    python3 tools/kstats.py $d/*/*kernel_stats.csv 40 | grep "ws64_kernel"
done
