#!/bin/bash
# HBM traffic of the convolution kernels and the scheduler update: separate rocprofv3 --pmc passes for FETCH_SIZE and
# WRITE_SIZE (they do not fit one pass), a few launches each.  Run on the GPU box from the repository root:
#   bash tools/pmc_traffic.sh gpurun_out/pmc_r02
# and then  python3 tools/pmc_traffic_json.py gpurun_out/pmc_r02  writes profiles/pmc_traffic.json.
set -e
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
export TMPDIR=/tmp ITERS=10
run() {  # name, counter, script args...
    local name=$1 ctr=$2; shift 2
    timeout -s KILL 120 rocprofv3 --pmc "$ctr" --output-format csv -d "$out/${name}_$ctr" -- python3 "$@" > "$out/${name}_$ctr.log" 2>&1
    echo "$name $ctr done" 
}
for ctr in FETCH_SIZE WRITE_SIZE; do
    run l3 $ctr tools/convone.py conv3x3 64 16 256 256
    run l2 $ctr tools/convone.py conv3x3 64 32 128 128
    run l1 $ctr tools/convone.py conv3x3 64 64 64 64
    run sched $ctr tools/schedone.py 64 64 64
    run rf $ctr tools/resone.py 64
    run ch $ctr tools/chainone.py 64 12
done
