"""Diagnostic: cycle stamps of the persistent 64-channel conv (needs a -DTAPCONV_STAMP build in LDM_HIP_LIB)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
os.environ["LDM_DBG_PTR"] = hex(dbg.data_ptr())
from tools.convbench import bench
bench("conv3x3", 64, 64, 64, 64, iters=20)
torch.cuda.synchronize()
v = dbg.cpu().tolist()
names = ["start", "weights in regs"] + [n for k in range(4) for n in (f"t{k} halo landed+barrier", f"t{k} next halo issued", f"t{k} compute done", f"t{k} staged+barrier", f"t{k} epilogue done")]
prev = v[0]
for i, n in enumerate(names):
    if v[i] == 0: break
    print(f"{i:2d} {n:28s} +{v[i]-prev:8d}   total {v[i]-v[0]:8d}")
    prev = v[i]
