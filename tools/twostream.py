"""Does the chip run two half-batch convolutions side by side faster than one full-batch launch?
Each arm is a graph of N launches; the two half-batch graphs are replayed on two streams at once."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffusionmodel_amd.ops import ConvOp

def mk(kind, B, S, Cin, Cout):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.05
    op = ConvOp(kind, w, b, dtype="bf16")
    x = torch.randn(B, S, S, Cin, device="cuda").to(torch.bfloat16)
    Ho, Wo = op.out_hw(S, S)
    out = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
    return op, x, out

def graph_of(op, x, out, n, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        op(x, out=out); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n):
                op(x, out=out)
    return g

def run(kind, S, C, Cout, N=40, reps=5):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    full = mk(kind, 64, S, C, Cout)
    h1, h2 = mk(kind, 32, S, C, Cout), mk(kind, 32, S, C, Cout)
    gf = graph_of(*full, N, s1)
    g1, g2 = graph_of(*h1, N, s1), graph_of(*h2, N, s2)
    torch.cuda.synchronize()
    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps / N * 1e3
    def one():
        with torch.cuda.stream(s1): gf.replay()
    def two():
        with torch.cuda.stream(s1): g1.replay()
        with torch.cuda.stream(s2): g2.replay()
    def half():
        with torch.cuda.stream(s1): g1.replay()
    print(f"{kind} {C}->{Cout} @{S}: B64 one stream {timed(one):6.1f} us | 2 x B32 two streams {timed(two):6.1f} us | B32 alone {timed(half):6.1f} us", flush=True)

run("conv3x3", 64, 64, 64)
run("conv3x3", 32, 128, 128)
run("conv3x3", 16, 256, 256)
run("conv3x3_s2", 64, 64, 128)
run("conv3x3_s2", 32, 128, 256)
