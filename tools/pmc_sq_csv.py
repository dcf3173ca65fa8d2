"""Per-kernel averages of the passes of tools/pmc_sq.sh as one CSV (argument 1: the passes' directory, argument 2: the
CSV to write).  Per kernel: launches, average duration of the kernel-trace pass, every counter's average per launch,
and the derived shares
  wait / stall / issue  = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint parts of a
                          wave's life: parked at s_waitcnt or a barrier, stalled at issue, issuing);
  mfma_busy             = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): share of the matrix
                          pipes' cycles with an MFMA in flight while the dispatch was active (GRBM_GUI_ACTIVE is summed
                          over the 8 XCDs and reads high on dispatches of a few tens of microseconds, so this is a lower
                          bound);
  mfma_busy_wall        = the same against 2.4 GHz x the traced duration (an un-profiled dispatch at the peak clock)."""
import collections, csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values
for p in ("a", "b"):
    for f in glob.glob(os.path.join(src, p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
names = sorted({c for k in ctr.values() for c in k})
def short(k):
    k = k.replace("void ldm::", "").replace("ldm::", "")
    for tail in ("(ldm::TapConvArgs)", "(ldm::AttnArgs)", "(ldm::SchedArgs)"):
        k = k.replace(tail, "")
    return k
rows = []
for k, cs in ctr.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    g = a.get("GRBM_GUI_ACTIVE", 0.0)
    calls, ns = dur.get(k, (0, 0.0))
    d = {"kernel": short(k), "launches": n, "trace_calls": calls, "trace_avg_us": round(ns / 1e3, 2)}
    d["wait"] = round(a.get("SQ_WAIT_ANY", 0) / wc, 3) if wc else ""
    d["stall"] = round(a.get("SQ_WAIT_INST_ANY", 0) / wc, 3) if wc else ""
    d["issue"] = round(a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3) if wc else ""
    d["mfma_busy"] = round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g / 8), 3) if g else ""
    d["mfma_busy_wall"] = round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * 2.4 * ns), 3) if ns else ""
    for c in names:
        d[c] = round(a[c], 1) if c in a else ""
    rows.append(d)
rows.sort(key=lambda d: -d["trace_avg_us"] * max(d["trace_calls"], 1))
with open(dst, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
for d in rows:
    print(f"{d['kernel'][:70]:70s} n={d['launches']:4d} {d['trace_avg_us']:8.2f} us  wait {d['wait']} stall {d['stall']} issue {d['issue']} "
          f"mfma_busy {d['mfma_busy']} (wall {d['mfma_busy_wall']})")
