#!/bin/bash
# SQ counters of every kernel of a UNet step (config 2 shape, eager launches): separate rocprofv3 --pmc passes, each
# within the 8 SQ + 2 GRBM slots of gfx950, plus one kernel-trace pass of the same program for the durations.
#   bash tools/pmc_sq.sh gpurun_out/pmc_sq   &&   python3 tools/pmc_sq_csv.py gpurun_out/pmc_sq profiles/rNN_pmc_sq_per_kernel.csv
out=${1:-gpurun_out/pmc_sq}
mkdir -p "$out"
export TMPDIR=/tmp
pass() {  # name, counters
    local name=$1; shift
    echo "pass $name: $*"
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/eagersteps.py 64 6 bf16 > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE || exit 1
timeout -s KILL 60 rocprofv3 -L > "$out/counters_available.txt" 2>&1 || true
pass b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE || echo "pass b skipped (a counter name is not known on this build)"
echo "pass trace"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/eagersteps.py 64 6 bf16 > "$out/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
echo "pmc_sq done"
