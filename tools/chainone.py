"""A few eager bf16 DDIM steps at [B,64,64,64]: target for rocprofv3 --pmc passes on the chained kernel (out_conv + DDIM
update + the next step's down1).  python3 tools/chainone.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffusionmodel_amd as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
m = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=A.make_synthetic_weights(A.LDMConfig(), 0))
z = m.ddim_sample(B, steps, spatial=64, seed=1, use_graph=False)
torch.cuda.synchronize()
print("done", float(z.abs().mean()), flush=True)
