"""Diagnostic: time the attention block (projections + core + output projection) and compare it with a
torch GPU statement of the same block.  usage: attnbench.py C B H W [iters]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffusionmodel_amd as A
from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes

C, B, H, W = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
w = synthetic_from_shapes(attention_param_shapes(C), 5)
torch.manual_seed(2)
x = (torch.randn(B, C, H, W) * 2.0).cuda()
m = A.Attention(C, dtype="bf16", weights=w)
print(f"attention C={C} B={B} {H}x{W}", flush=True)
got = m(x)
torch.cuda.synchronize()
# torch statement (f32, one image at a time): names as in weights.attention_param_shapes
names = list(w.keys())
print("params:", names, flush=True)
wd = {k: v.cuda().float() for k, v in w.items()}
def lin(t, name):
    return t @ wd[name + ".weight"] + wd[name + ".bias"] if name + ".weight" in wd else None
errs = []
for i in range(0, B, max(1, B // 4)):
    t = x[i].reshape(C, H * W).t()
    q, k, v = lin(t, "question"), lin(t, "key"), lin(t, "val")
    if q is None:
        print("unknown parameter names; skipping the reference"); break
    p = torch.softmax(q @ k.t() / C ** 0.5, dim=-1)
    y = lin(p @ v, "out") + t
    ref = y.t().reshape(C, H, W)
    d = (got[i].float() - x[i]) - (ref - x[i])
    errs.append((d.norm() / (ref - x[i]).norm()).item())
print("branch rel_err per sampled image:", ["%.3e" % e for e in errs], flush=True)
for _ in range(2): m(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): m(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
N = H * W
fl = B * (4.0 * N * N * C)
print(f"block {ms*1e3:.1f} us per call; core flops {fl/1e12:.3f} T -> {fl/ms/1e9:.0f} TFLOP/s if the core were all of it", flush=True)
