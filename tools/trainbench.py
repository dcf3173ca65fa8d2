"""Training-mode forward of the UNet (batch-statistics BatchNorm, dropout, one timestep per image) against the
inference forward on the same input: time per call, events on the current stream.
  python3 tools/trainbench.py [B] [S] [dtype]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffusionmodel_amd as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
cfg = A.LDMConfig()
m = A.UNet(64, 256, dtype=dtype, weights=A.make_synthetic_weights(cfg, 0))
x = torch.randn(B, 64, S, S, device="cuda")
te = A.sin_time_addition(torch.randint(0, 1000, (B,), device="cuda"), 256)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ev = timed(lambda: m.forward(x, te))
tr = timed(lambda: m.forward(x, te, train=True, dropout_seed=1))
print(f"UNet forward [{B},64,{S},{S}] {dtype}: inference {ev:.0f} us, training mode {tr:.0f} us ({tr / ev:.2f} x)", flush=True)
