"""Developer check of one conv shape against torch (f32 on the GPU): relative L2 error.
usage: convcheck.py kind B S Cin Cout"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffusionmodel_amd.ops import ConvOp
kind, B, S, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
g = torch.Generator().manual_seed(0)
res = len(sys.argv) > 6 and sys.argv[6] == "res"
if kind == "convT4x4_s2":
    w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.05
elif kind == "conv1x1":
    w = torch.randn(Cout, Cin, generator=g) * 0.05
else:
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
b = torch.randn(Cout, generator=g) * 0.05
op = ConvOp(kind, w, b, dtype="bf16")
x = torch.randn(B, S, S, Cin, device="cuda").to(torch.bfloat16)
r = torch.randn(B, S, S, Cout, device="cuda").to(torch.bfloat16) if res else None
y = op(x, res=r).float()
xr = x.float().permute(0, 3, 1, 2)
wb = w.to(torch.bfloat16).float().cuda()
if kind == "conv1x1":
    ref = F.conv2d(xr, wb[:, :, None, None], b.cuda())
elif kind == "conv3x3":
    ref = F.conv2d(xr, wb, b.cuda(), padding=1)
elif kind == "conv3x3_s2":
    ref = F.conv2d(xr, wb, b.cuda(), padding=1, stride=2)
else:
    ref = F.conv_transpose2d(xr, wb, b.cuda(), stride=2, padding=1)
ref = ref.permute(0, 2, 3, 1)
if res:
    ref = ref + r.float()
err = ((y - ref).norm() / ref.norm()).item()
print(f"{os.environ.get('LDM_HIP_LIB', 'default')[-20:]} {kind} B{B} S{S} {Cin}->{Cout}: rel_err {err:.3e}", flush=True)
assert err < 1e-2
