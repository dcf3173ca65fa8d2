"""Diagnostic: where a workgroup of a weight-stationary convolution kernel (3x3 stride 1 at 64 channels, stride 2) spends its life (needs a -DTAPCONV_STAMP build
in LDM_HIP_LIB): start -> weights and first window landed -> end of each of the first four tiles -> end.
usage: wslife.py kind B S Cin Cout"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rt = torch.zeros(8 * 1024, dtype=torch.int64, device="cuda")
os.environ["LDM_DBG_RT_PTR"] = hex(rt.data_ptr())
from tools.convbench import bench
kind, B, S, Cin, Cout = sys.argv[1], *(int(x) for x in sys.argv[2:6])
bench(kind, B, S, Cin, Cout, iters=50)
torch.cuda.synchronize()
v = rt.cpu().view(-1, 8)
v = v[v[:, 6] > 0].double() / 100.0
t0 = v[:, 0].min()
names = ["start", "prologue done", "tile 0", "tile 1", "tile 2", "tile 3", "end"]
print(f"{v.shape[0]} workgroups; microseconds after the first workgroup's start (min / median / max)")
for k, n in enumerate(names):
    c = v[:, k][v[:, k] > 0] - t0
    if c.numel():
        qs = torch.quantile(c, torch.tensor([0.1, 0.25, 0.75, 0.9, 0.99], dtype=c.dtype))
        print(f"  {n:14s} {float(c.min()):6.2f} {float(c.median()):6.2f} {float(c.max()):6.2f}   p10/25/75/90/99 " + " ".join(f"{float(x):6.2f}" for x in qs))
