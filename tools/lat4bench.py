"""Throughput of the sampler at the reference's literal latent [B,64,4,4]: python tools/lat4bench.py [B ...]
Set LDM_DENSE4=0 to time the tap kernels at 4x4 instead of the whole-image matrices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusionmodel_amd as A

def run(B, dtype, mode, reps=5):
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    m = A.DiffusionModel(64, 3, 50 if mode == "ddpm" else 1000, 256, dtype=dtype, weights=w)
    f = (lambda: m.denoising_process(B, spatial=4, seed=1)) if mode == "ddpm" else (lambda: m.ddim_sample(B, 50, spatial=4, seed=1))
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"dense4={os.environ.get('LDM_DENSE4','1')} B={B} {dtype} {mode}: {dt*1e3:.2f} ms per 50-step call, {dt/50*1e6:.1f} us/step, {B/dt:.0f} images/s", flush=True)

for B in [int(a) for a in sys.argv[1:]] or [4, 64, 1024]:
    run(B, "bf16", "ddim")
    run(B, "fp32", "ddpm")
