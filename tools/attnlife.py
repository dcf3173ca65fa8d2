"""Diagnostic: where a workgroup of the streaming attention kernel spends its life (needs a -DATTN_STAMP build of
attention.hip in LDM_HIP_LIB): start -> first tiles landed -> first three key/value tiles -> all key/value tiles -> end
(with the output projection).  usage: attnlife.py B H W   (256 channels, bf16)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rt = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
os.environ["LDM_DBG_RT_PTR"] = hex(rt.data_ptr())
import diffusionmodel_amd as A
from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes
B, H, W = (int(x) for x in sys.argv[1:4])
att = A.Attention(256, dtype="bf16", weights=synthetic_from_shapes(attention_param_shapes(256), 5))
x = torch.randn(B, 256, H, W, device="cuda")
for _ in range(20): y = att(x)
torch.cuda.synchronize()
v = rt.cpu().view(-1, 8)
v = v[v[:, 6] > 0].double() / 100.0
t0 = v[:, 0].min()
names = ["start", "tiles landed", "kv tile 0", "kv tile 1", "kv tile 2", "kv tiles done", "end"]
print(f"{v.shape[0]} workgroups; microseconds after the first workgroup's start (min / median / max)")
for k, n in enumerate(names):
    c = v[:, k][v[:, k] > 0] - t0
    if c.numel(): print(f"  {n:14s} {float(c.min()):6.2f} {float(c.median()):6.2f} {float(c.max()):6.2f}")
