"""Training-mode BatchNorm kernels on the UNet's activation shapes: time and achieved HBM rate of the statistics pass
(one read) and of the normalise + SiLU (+ time vector | + residual) pass (read(s) + write).  Every variant is
captured into a graph of N launches and replayed, so host launch cost is not in the numbers.
  python3 tools/bnbench.py [B]        (on the GPU box, from the repository root)"""
import ctypes as C
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusionmodel_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _lib.load()
N = 20


def graph_time(fn):
    """microseconds per call of fn (launches on the current stream), from a replayed graph of N calls"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(N):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * N) * 1e3


p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())   # noqa: E731
for dtype, td, es, code in (("bf16", torch.bfloat16, 2, _lib.LDM_BF16), ("fp32", torch.float32, 4, _lib.LDM_F32)):
    for Cc, S in ((64, 64), (128, 32), (256, 16)):
        g = torch.Generator(device="cuda").manual_seed(1)
        h = torch.randn(B, S, S, Cc, device="cuda", generator=g).to(td)
        res = torch.randn(B, S, S, Cc, device="cuda", generator=g).to(td)
        tv = torch.randn(B, Cc, device="cuda", generator=g)
        gamma, beta = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        out = torch.empty_like(h)
        ws = torch.empty(lib.ldm_batchnorm_ws_floats(code, B * S * S, Cc), device="cuda")

        def call(o, r, t):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(lib.ldm_batchnorm_act(code, p(h), p(r), p(o), B, S * S, Cc, p(gamma), p(beta), 1e-5, 1, p(t), 0.0, 0,
                                             p(ws), None, None, stream))
        nbytes = h.numel() * es
        t_stats = graph_time(lambda: call(None, None, None))
        t_tv = graph_time(lambda: call(out, None, tv)) - t_stats
        t_res = graph_time(lambda: call(out, res, None)) - t_stats
        print(f"{dtype} [{B},{S},{S},{Cc}] {nbytes/1e6:6.1f} MB: statistics (2 launches) {t_stats:6.1f} us = {nbytes/t_stats/1e6:5.2f} TB/s | "
              f"normalise+SiLU+t {t_tv:6.1f} us = {2*nbytes/t_tv/1e6:5.2f} TB/s | normalise+residual+SiLU {t_res:6.1f} us = "
              f"{3*nbytes/t_res/1e6:5.2f} TB/s", flush=True)
