"""Diagnostic: one 64-chain DDIM call against two 32-chain calls replayed on two streams at once."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffusionmodel_amd as A
cfg = A.LDMConfig()
w = A.make_synthetic_weights(cfg, 0)
full = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
ha = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
hb = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
z64 = torch.randn(64, 64, 64, 64, device="cuda")
za, zb = z64[:32].contiguous(), z64[32:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    full.ddim_sample(64, 50, spatial=64, z0=z64)
def two():
    with torch.cuda.stream(s1):
        ha.ddim_sample(32, 50, spatial=64, z0=za)
    with torch.cuda.stream(s2):
        hb.ddim_sample(32, 50, spatial=64, z0=zb, chain0=32)
def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print(f"one 64-chain call: {timed(one):.2f} ms", flush=True)
print(f"two 32-chain calls on two streams: {timed(two):.2f} ms", flush=True)
def half():
    with torch.cuda.stream(s1):
        ha.ddim_sample(32, 50, spatial=64, z0=za)
print(f"one 32-chain call alone: {timed(half):.2f} ms", flush=True)
