"""Diagnostic: time add_noise and the loss reduction on [B,64,S,S] f32.  usage: noisebench.py [B] [S]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import diffusionmodel_amd as A
from diffusionmodel_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=A.make_synthetic_weights(A.LDMConfig(), 0))
x, n = torch.randn(B, 64, S, S, device="cuda"), torch.randn(B, 64, S, S, device="cuda")
t = torch.randint(0, 1000, (B,), device="cuda")
def timed(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
ms = timed(lambda: m.add_noise(x, t, n))
nb = x.numel() * 4 * 3
print(f"add_noise [{B},64,{S},{S}]: {ms*1e3:.1f} us, {nb/1e6:.0f} MB -> {nb/ms/1e9:.2f} TB/s", flush=True)
lib = _lib.load()
ws, out = torch.empty(1024, device="cuda"), torch.empty(1, device="cuda")
p = lambda v: C.c_void_p(v.data_ptr())
ms = timed(lambda: _lib.check(lib.ldm_mse(p(x), p(n), x.numel(), p(ws), p(out), C.c_void_p(torch.cuda.current_stream().cuda_stream))))
nb = x.numel() * 4 * 2
print(f"mse: {ms*1e3:.1f} us, {nb/1e6:.0f} MB -> {nb/ms/1e9:.2f} TB/s", flush=True)
