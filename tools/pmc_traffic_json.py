"""Summarise the passes of tools/pmc_traffic.sh into profiles/pmc_traffic.json: one record per (kernel, shape) with
HBM bytes per launch = 2 x FETCH_SIZE (gfx950 tallies 128-byte read requests as 64 bytes) + WRITE_SIZE, both in KB in
the counter output."""
import csv, glob, json, os, sys
src = sys.argv[1]
SHAPES = {"l3": ("3x3 256->256 on [64,16,16,256] bf16", 2 * 64 * 16 * 16 * 256 * 2 + 256 * 256 * 9 * 2),
          "l2": ("3x3 128->128 on [64,32,32,128] bf16", 2 * 64 * 32 * 32 * 128 * 2 + 128 * 128 * 9 * 2),
          "l1": ("3x3 64->64 on [64,64,64,64] bf16", 2 * 64 * 64 * 64 * 64 * 2 + 64 * 64 * 9 * 2),
          "sched": ("[64,64,64,64] f32 ddpm", 16 * 64 * 64 * 64 * 64),
          # the fused ResidualBlock: x in, block output out, two weight sets (its forward also converts NCHW <-> NHWC
          # around the kernel; only the kernel's own launches are averaged)
          "rf": ("ResidualBlock 64 channels on [64,64,64,64] bf16", 2 * 64 * 64 * 64 * 64 * 2 + 2 * 64 * 64 * 9 * 2),
          # the chained DDIM step: h in, f32 state in and out, down1's output out, two weight sets
          "ch": ("out_conv + DDIM update + down1 on [64,64,64,64] bf16", 64 * 64 * 64 * 64 * (2 + 4 + 4 + 2) + 2 * 64 * 64 * 9 * 2)}
WANT = {"l3": "tapconv", "l2": "tapconv", "l1": "tapconv", "sched": "sched_step_kernel", "rf": "resfuse64_kernel<0>",
        "ch": "resfuse64_kernel<1>"}
def avg(name, ctr):
    files = glob.glob(os.path.join(src, f"{name}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    assert files, (name, ctr)
    per = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == ctr and WANT[name] in r["Kernel_Name"]:
            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    (kn, vals), = per.items()
    vals = vals[3:] if len(vals) > 6 else vals          # skip the first launches (cold caches, weights from HBM)
    return kn, sum(vals) / len(vals), len(vals)
records = []
for name, (shape, alg) in SHAPES.items():
    kn, fetch_kb, n = avg(name, "FETCH_SIZE")
    kn2, write_kb, _ = avg(name, "WRITE_SIZE")
    assert kn == kn2
    key = kn if name != "sched" else "ldm::sched_step_kernel"
    records.append({"kernel": key, "kernel_as_profiled": kn, "shape": shape,
                    "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
                    "FETCH_SIZE_KB_avg": fetch_kb, "WRITE_SIZE_KB_avg": write_kb, "launches": n,
                    "algorithmic_bytes": alg})
out = {"correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE exact for "
                     "16-B-per-lane stores",
       "command": "bash tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE, python3 tools/convone.py / "
                  "tools/schedone.py)",
       "records": records}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"), "w"), indent=1)
for r in records:
    print(f"{r['shape']:45s} {r['hbm_bytes_per_launch']/1e6:8.1f} MB (algorithmic {r['algorithmic_bytes']/1e6:.1f})  {r['kernel'][:90]}")
