"""The DDPM scheduler update on [B,C,S,S] f32 with the noise read from memory, a few launches: target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffusionmodel_amd as A
from diffusionmodel_amd import schedule
B, C, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
z = torch.randn(B, C, S, S, device="cuda"); eps = torch.randn_like(z); noise = torch.randn_like(z)
row = schedule.ddpm_coefficients(1000)[500]
for _ in range(int(os.environ.get("ITERS", "10"))):
    A.sched_step(z, eps, row, "ddpm", last=False, noise=noise)
torch.cuda.synchronize()
