"""Diagnostic: lifetime of every workgroup of the generic conv kernel against the kernel's duration (needs a
-DTAPCONV_STAMP build in LDM_HIP_LIB): how much of a launch is dispatch ramp and drain, i.e. what a kernel that never
ends between layers could remove at most.  usage: wglife.py kind B S Cin Cout"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rt = torch.zeros(8192, dtype=torch.int64, device="cuda")
os.environ["LDM_DBG_RT_PTR"] = hex(rt.data_ptr())
from tools.convbench import bench
kind, B, S, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
bench(kind, B, S, Cin, Cout, iters=50)
torch.cuda.synchronize()
v = rt.cpu().view(-1, 2)
v = v[v[:, 1] > 0].double() / 100.0           # microseconds (100-MHz counter); the last launch's stamps
t0 = v[:, 0].min()
life = v[:, 1] - v[:, 0]
print(f"{v.shape[0]} workgroups: first start 0, last start {float(v[:,0].max()-t0):.2f} us; first end {float(v[:,1].min()-t0):.2f}, "
      f"last end {float(v[:,1].max()-t0):.2f} us; lifetime min {float(life.min()):.2f} median {float(life.median()):.2f} max {float(life.max()):.2f} us")
