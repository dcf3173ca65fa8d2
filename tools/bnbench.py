"""Training-mode BatchNorm kernels on the UNet's activation shapes: time and achieved HBM rate of the statistics pass
(one read) and of the normalise + SiLU (+ time vector | + residual) pass (read(s) + write).
  python3 tools/bnbench.py [B]        (on the GPU box, from the repository root)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusionmodel_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


for dtype, td, es in (("bf16", torch.bfloat16, 2), ("fp32", torch.float32, 4)):
    for C, S in ((64, 64), (128, 32), (256, 16)):
        g = torch.Generator(device="cuda").manual_seed(1)
        h = torch.randn(B, S, S, C, device="cuda", generator=g).to(td)
        res = torch.randn(B, S, S, C, device="cuda", generator=g).to(td)
        tv = torch.randn(B, C, device="cuda", generator=g)
        gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        out = torch.empty_like(h)
        nbytes = h.numel() * es
        t_stats = timeit(lambda: ops.batchnorm_act(h, gamma, beta, stats_only=True))
        t_tv = timeit(lambda: ops.batchnorm_act(h, gamma, beta, act="silu", tvec=tv, out=out)) - t_stats
        t_res = timeit(lambda: ops.batchnorm_act(h, gamma, beta, act="silu", res=res, out=out)) - t_stats
        print(f"{dtype} [{B},{S},{S},{C}] {nbytes/1e6:6.1f} MB: statistics {t_stats:6.1f} us = {nbytes/t_stats/1e6:5.2f} TB/s | "
              f"normalise+SiLU+t {t_tv:6.1f} us = {2*nbytes/t_tv/1e6:5.2f} TB/s | normalise+residual+SiLU {t_res:6.1f} us = "
              f"{3*nbytes/t_res/1e6:5.2f} TB/s", flush=True)
