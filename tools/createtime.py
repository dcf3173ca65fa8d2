import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, diffusionmodel_amd as A
for cfg, name in ((A.LDMConfig(), "default"), (A.LDMConfig.scaled(), "scaled")):
    w = A.make_synthetic_weights(cfg, 0)
    for dt in ("bf16", "fp32"):
        t0 = time.perf_counter()
        m = A.UNet(cfg.in_channels, cfg.time_embed_dim, cfg.channels, cfg.scaled_attention, dt, w)
        torch.cuda.synchronize()
        print(name, dt, f"UNet create {time.perf_counter()-t0:.2f}s", flush=True)
        del m
