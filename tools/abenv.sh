#!/bin/bash
# A/B of one environment switch of the library on the default benchmark, alternating inside one GPU call:
#   tools/abenv.sh VAR [rounds] [extra bench.py flags...]     runs bench.py with VAR=0 and VAR=1 in turn
# and appends "VAR=v images/s ms_per_call" lines to gpurun_out/abenv_VAR.txt.
set -e
var=$1; rounds=${2:-3}; shift; shift || true
out=gpurun_out/abenv_${var}.txt
mkdir -p gpurun_out; : > "$out"
for i in $(seq "$rounds"); do
  for v in 0 1; do
    line=$(env "$var=$v" timeout -k 10 200 python bench.py --steps 10 --warmup 3 "$@" 2>/dev/null |
           python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
    echo "$var=$v $line" >> "$out"
  done
done
cat "$out"
