"""Diagnostic: per-step cycle stamps of the generic conv kernel (needs a -DTAPCONV_STAMP build in LDM_HIP_LIB).
usage: stamps.py kind B S Cin Cout"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dbg = torch.zeros(512, dtype=torch.int64, device="cuda")
os.environ["LDM_DBG_PTR"] = hex(dbg.data_ptr())
from tools.convbench import bench
kind, B, S, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
bench(kind, B, S, Cin, Cout, iters=20)
torch.cuda.synchronize()
v = dbg.cpu().tolist()
for w, base in (("wave 0", 0), ("last wave", 256)):
    s = v[base:base + 250]
    n = max(i for i, x in enumerate(s) if x) + 1
    t0 = s[0]
    print(f"{w}: {n} stamps, total {s[n-1]-t0} cycles; prologue (start -> first step) {s[1]-t0}")
    steps = (n - 3) // 3
    tot = dict(compute=0, vmwait=0, lgkm=0, barrier=0)
    for k in range(steps):
        A, Bv, C = s[1 + 3 * k], s[2 + 3 * k], s[3 + 3 * k]
        nxt = s[1 + 3 * (k + 1)] if k + 1 < steps else s[1 + 3 * steps]
        row = (Bv - A, C - Bv, nxt - C)
        if k < 12 or k >= steps - 3:
            print(f"  step {k:2d}: issue + batch A + LDS reads done {row[0]:6d} | vmcnt wait {row[1]:5d} | barrier + batch B {row[2]:6d}")
        tot["compute"] += row[0]; tot["lgkm"] += row[1]; tot["barrier"] += row[2]
    print(f"  sums over {steps} steps: first part {tot['compute']}, vmcnt wait {tot['lgkm']}, barrier + batch B {tot['barrier']}")
    print(f"  epilogue (last barrier -> end) {s[n-1]-s[n-2]}")
