"""The ResidualBlock of the 64-channel level at [B,64,64,64], bf16, a few forward calls: target for rocprofv3 --pmc
passes (LDM_RESFUSE=0: the two convolution launches instead of the fused kernel).  python3 tools/resone.py [B]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffusionmodel_amd as A
from diffusionmodel_amd.weights import resblock_param_shapes, synthetic_from_shapes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = A.ResiduelBlock(64, 256, dtype="bf16", weights=synthetic_from_shapes(resblock_param_shapes(64, 256), 1))
x = torch.randn(B, 64, 64, 64, device="cuda")
t = torch.randn(B, 256, device="cuda")
for _ in range(int(os.environ.get("ITERS", "10"))):
    y = m(x, t)
torch.cuda.synchronize()
print("done", float(y.abs().mean()), flush=True)
