"""Diagnostic: the life of the fused ResidualBlock kernel's workgroups (needs a -DTAPCONV_STAMP build in LDM_HIP_LIB):
start -> prologue done -> end of each step -> end, for the producer wave 0 and the consumer wave 4 of every workgroup.
usage: rflife.py [B] [chain]      "chain": the chained DDIM kernel (out_conv + update + next down1) of a two-step DDIM
run instead of the ResidualBlock (LDM_DBG_RT_ONLY=chain keeps the other stamping kernels off the buffer)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rt = torch.zeros(B * 4 * 2 * 16, dtype=torch.int64, device="cuda")
os.environ["LDM_DBG_RT_PTR"] = hex(rt.data_ptr())
import diffusionmodel_amd as A
from diffusionmodel_amd.weights import resblock_param_shapes, synthetic_from_shapes
if len(sys.argv) > 2 and sys.argv[2] == "chain":
    os.environ["LDM_DBG_RT_ONLY"] = "chain"
    dm = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=A.make_synthetic_weights(A.LDMConfig(), 0))
    for _ in range(3):
        dm.ddim_sample(B, 2, spatial=64, seed=1, use_graph=False)
else:
    m = A.ResiduelBlock(64, 256, dtype="bf16", weights=synthetic_from_shapes(resblock_param_shapes(64, 256), 1))
    x = torch.randn(B, 64, 64, 64, device="cuda")
    t = torch.randn(B, 256, device="cuda")
    for _ in range(5):
        m(x, t)
torch.cuda.synchronize()
v = rt.cpu().view(-1, 2, 16).double() / 100.0
t0 = v[:, :, 0].min()
names = ["start", "prologue"] + [f"step {k}" for k in range(11)] + ["end"]
for role, rn in enumerate(("producer wave 0", "consumer wave 4")):
    print(rn, "(microseconds after the first start: min / median / max; median step length)")
    prev = None
    for k, n in enumerate(names):
        c = v[:, role, k] - t0
        d = "" if prev is None else f"   +{float((v[:, role, k] - prev).median()):5.2f}"
        print(f"  {n:10s} {float(c.min()):6.2f} {float(c.median()):6.2f} {float(c.max()):6.2f}{d}")
        prev = v[:, role, k]
