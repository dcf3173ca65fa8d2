"""Chained DDIM steps against the single-step kernels: where do the final latents differ?  python tools/chaindiff.py B H steps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusionmodel_amd as A
B, H, steps = (int(x) for x in sys.argv[1:4])
w = A.make_synthetic_weights(A.LDMConfig(), 0)
z0 = torch.randn(B, 64, H, 64, generator=torch.Generator().manual_seed(5)).cuda()
os.environ["LDM_DDIMCHAIN"] = "0"
m0 = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
plain = m0.ddim_sample(B, steps, spatial=(H, 64), z0=z0, use_graph=False)
del os.environ["LDM_DDIMCHAIN"]
m1 = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
ch = m1.ddim_sample(B, steps, spatial=(H, 64), z0=z0, use_graph=False)
torch.cuda.synchronize()
d = (ch - plain).abs()
print("max abs diff", d.max().item(), "rel", (d.norm() / plain.norm()).item(), "differing", (d > 0).float().mean().item())
bad = (d > 0)
print("by row   ", bad.float().mean(dim=(0, 1, 3)).cpu().numpy().round(3))
print("by column", bad.float().mean(dim=(0, 1, 2)).cpu().numpy().round(3))
print("by chan  ", bad.float().mean(dim=(0, 2, 3)).cpu().numpy().round(3))
print("by image ", bad.float().mean(dim=(1, 2, 3)).cpu().numpy().round(3))
