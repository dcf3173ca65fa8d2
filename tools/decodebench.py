"""Diagnostic: time Vae::decode + the 8-bit conversion on [B,64,4,4] latents.  usage: decodebench.py [B] [dtype] [iters]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffusionmodel_amd as A

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dec = A.Decoder(3, 64, dtype=dtype, weights=A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 21))
z = torch.randn(B, 64, 4, 4, device="cuda")
print(f"decode B={B} {dtype}", flush=True)
for _ in range(3):
    rgb = A.to_rgb8(dec(z))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    rgb = A.to_rgb8(dec(z))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
wbytes = 1024 * 32768 * (2 if dtype == "bf16" else 4)
print(f"decode + rgb8: {ms*1e3:.1f} us per batch of {B} = {B/ms*1e3:.0f} images/s; the Linear's weights alone are {wbytes/1e6:.0f} MB", flush=True)
