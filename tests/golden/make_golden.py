#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU restatement (oracle/).

The reference ships no fixtures and cannot be run (its Python file does not parse, its Rust path
needs cargo + burn + a Vulkan GPU and fails on a shape mismatch), so these vectors come from this
repository's own restatement of src/main.rs; they pin it against drift and give the GPU tests
inputs and expected outputs that do not need the oracle at run time.  Weights are not stored: they
are regenerated from seeds by diffusionmodel_amd.weights (seeded torch CPU generators).

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from diffusionmodel_amd.weights import (LDMConfig, attention_param_shapes, decoder_param_shapes,  # noqa: E402
                                        make_synthetic_weights, resblock_param_shapes, synthetic_from_shapes)
from oracle import ldm_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def main():
    torch.set_num_threads(1)     # fixed reduction order
    # schedule and embedding
    np.savez(os.path.join(OUT, "schedule.npz"),
             betas50=O.get_betas(50), abar50=O.get_alphas(50), ddpm50=O.ddpm_coefficients(50),
             betas1000=O.get_betas(1000), abar1000=O.get_alphas(1000), ddpm1000=O.ddpm_coefficients(1000),
             ddim_ts=np.array(O.ddim_timesteps(1000, 50), dtype=np.int32), ddim=O.ddim_coefficients(1000, 50),
             sin_t=np.array([0, 1, 49, 999], dtype=np.int32),
             sin_emb=O.sin_time_addition(torch.tensor([0, 1, 49, 999]), 256).numpy())
    # operators
    ops = {}
    w = synthetic_from_shapes(resblock_param_shapes(64, 256), 11)
    x, t = randn((2, 64, 4, 4), 1), randn((2, 256), 2)
    ops.update(res_x=x.numpy(), res_t=t.numpy(),
               res_y=O.residual_block({"m." + k: v for k, v in w.items()}, "m", x, t).numpy())
    w = synthetic_from_shapes(attention_param_shapes(128), 5)
    x = randn((2, 128, 4, 4), 3) * 2.0
    ops.update(att_x=x.numpy(), att_y=O.attention({"m." + k: v for k, v in w.items()}, "m", x).numpy())
    np.savez(os.path.join(OUT, "ops.npz"), **ops)
    # UNet forward
    cfg = LDMConfig()
    w = make_synthetic_weights(cfg, 0)
    un = {}
    for name, (B, S, ts) in {"lat4": (4, 4, [7, 31, 0, 49]), "s16": (2, 16, [900, 3])}.items():
        x = randn((B, 64, S, S), 3)
        te = O.sin_time_addition(torch.tensor(ts), 256)
        un[name + "_x"] = x.numpy()
        un[name + "_t"] = np.array(ts, dtype=np.int32)
        un[name + "_eps"] = O.unet_forward(w, x, te).numpy()
    np.savez(os.path.join(OUT, "unet.npz"), **un)
    # sampling chains (z0 / noise are regenerated from these seeds by the tests)
    T, B, S = 50, 4, 4
    z0, noise = randn((B, 64, S, S), 1234), randn((T, B, 64, S, S), 4321)
    z, zs, eps = O.denoising_process(w, z0, T, noise=noise, trace=True)
    ch = dict(ddpm_final=z.numpy(), ddpm_eps0=eps[0].numpy(), ddpm_eps25=eps[25].numpy(), ddpm_eps49=eps[49].numpy(),
              ddpm_z25=zs[25].numpy())
    z0 = randn((2, 64, 8, 8), 7)
    z, zs, eps = O.denoising_process(w, z0, 1000, mode="ddim", ddim_steps=50, trace=True)
    ch.update(ddim_final=z.numpy(), ddim_eps0=eps[0].numpy(), ddim_z10=zs[10].numpy())
    np.savez(os.path.join(OUT, "chains.npz"), **ch)
    # VAE decoder and the 8-bit image (decoder weights: seed 21)
    wd = synthetic_from_shapes(decoder_param_shapes(3, 64), 21)
    zd = randn((2, 64, 4, 4), 55)
    img = O.vae_decode(wd, "", zd)
    np.savez(os.path.join(OUT, "decode.npz"), img=img.numpy(), rgb=O.to_rgb8(img.numpy()))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
