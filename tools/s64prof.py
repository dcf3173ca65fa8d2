"""One graph-replayed 50-step DDIM call at [B,64,64,64] for a kernel trace: python tools/s64prof.py B dtype"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusionmodel_amd as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
m = A.DiffusionModel(64, 3, 1000, 256, dtype=dtype, weights=A.make_synthetic_weights(A.LDMConfig(), 0))
for _ in range(3):
    m.ddim_sample(B, 50, spatial=64, seed=1)
torch.cuda.synchronize()
