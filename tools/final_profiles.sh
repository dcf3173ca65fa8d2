#!/bin/bash
# Benchmark lines and rocprofv3 summaries of every benchmark configuration: run on the GPU box from the repository root,
#   bash tools/final_profiles.sh gpurun_out/final TAG
# then copy the files into profiles/ with the tag as prefix.
set -e
out=${1:-gpurun_out/final}; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_under_profiler.json 2> $out/bench_prof.err
timeout -k 10 300 python3 bench.py --sampler ddpm --ddim-steps 1000 --dtype fp32 --steps 2 --warmup 1 --no-cpu-baseline > $out/config5.json 2> $out/config5.err
timeout -k 10 300 python3 bench.py --sampler ddpm --ddim-steps 50 --num-time 50 --dtype fp32 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > $out/config1.json 2> $out/config1.err
timeout -k 10 300 python3 bench.py --spatial 4 --batch 4 --steps 20 --warmup 3 --no-cpu-baseline > $out/config1_lat4.json 2> $out/config1_lat4.err
timeout -k 10 600 python3 bench.py --scaled --spatial 256 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline > $out/config4_b16.json 2> $out/config4_b16.err
timeout -k 10 900 python3 bench.py --scaled --spatial 256 --batch 128 --steps 1 --warmup 1 --no-cpu-baseline > $out/config4_b128.json 2> $out/config4_b128.err
for b in 32 128 256; do timeout -k 10 300 python3 bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline > $out/config2_b$b.json 2> $out/config2_b$b.err; done
bash tools/pmc_traffic.sh $out/pmc > $out/pmc.log 2>&1
python3 tools/pmc_traffic_json.py $out/pmc > $out/pmc_summary.txt
cp profiles/pmc_traffic.json $out/pmc_traffic.json
bash tools/pmc_sq.sh $out/pmc_sq > $out/pmc_sq.log 2>&1
python3 tools/pmc_sq_csv.py $out/pmc_sq $out/pmc_sq_per_kernel.csv > $out/pmc_sq_summary.txt
timeout -k 10 200 python3 tools/bnbench.py 64 > $out/bnbench.txt 2>&1
for f in $out/*.json; do [ "$(basename $f)" = pmc_traffic.json ] && continue; python3 - $f <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], round(r["value"], 1), r["unit"], "ms/step", round(r["ms_per_step"], 2), "roofline", r.get("roofline", {}).get("bound"), round(r.get("roofline", {}).get("frac", 0), 3))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
