"""A few eager (ungraphed) DDIM steps at [B,64,64,64]: every kernel of a UNet step as its own dispatch, the target of
the rocprofv3 --pmc passes of tools/pmc_sq.sh.   python3 tools/eagersteps.py [B] [steps] [dtype]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffusionmodel_amd as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
m = A.DiffusionModel(64, 3, 1000, 256, dtype=dtype, weights=A.make_synthetic_weights(A.LDMConfig(), 0))
z = m.ddim_sample(B, steps, spatial=64, seed=1, use_graph=False)
torch.cuda.synchronize()
print("done", float(z.abs().mean()), flush=True)
