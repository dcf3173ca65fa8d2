"""Per-step teacher-forced epsilon error of a 50-step chain at [B,64,S,S] (developer diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusionmodel_amd as A
from oracle import ldm_oracle as O
from tests.util import rel_err

B, S, steps, T = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 50, 1000
w = A.make_synthetic_weights(A.LDMConfig(), 0)
z0 = torch.randn(B, 64, S, S, generator=torch.Generator().manual_seed(1234))
ref_z, zs, eps = O.denoising_process(w, z0, T, mode="ddim", ddim_steps=steps, trace=True)
ts = O.ddim_timesteps(T, steps)
mb = A.DiffusionModel(64, 3, T, 256, dtype="bf16", weights=w)
mf = A.DiffusionModel(64, 3, T, 256, dtype="fp32", weights=w)
wb = {k: v.bfloat16().float() for k, v in w.items()}
for k, t in enumerate(ts):
    temb = O.sin_time_addition(torch.full((B,), t), 256)
    eb = mb.noise_predict(zs[k].cuda(), temb.cuda())
    ef = mf.noise_predict(zs[k].cuda(), temb.cuda())
    # the oracle on bf16-rounded weights and input: how much of the error is the rounding of the operands alone
    er = O.unet_forward(wb, zs[k].bfloat16().float(), temb)
    print(f"k={k:2d} t={t:3d} rms z={zs[k].pow(2).mean().sqrt():.3f} eps={eps[k].pow(2).mean().sqrt():.4f} "
          f"bf16={rel_err(eb, eps[k]):.3e} f32={rel_err(ef, eps[k]):.3e} operands-rounded oracle={rel_err(er, eps[k]):.3e}", flush=True)
