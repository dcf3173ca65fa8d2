"""Diagnostic: how much of the in-graph slowdown of a convolution is cold L2 -- rotate the weights, the inputs, or
both between launches of the same shape."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffusionmodel_amd.ops import ConvOp
B, S, C = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 256
NW, NX = 16, 16
g = torch.Generator().manual_seed(0)
ops = [ConvOp("conv3x3", torch.randn(C, C, 3, 3, generator=g) * 0.02, torch.zeros(C), dtype="bf16") for _ in range(NW)]
xs = [torch.randn(B, S, S, C, device="cuda").bfloat16() for _ in range(NX)]
out = torch.empty(B, S, S, C, device="cuda", dtype=torch.bfloat16)
def run(rot_w, rot_x, iters=320):
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops[0](xs[0], out=out); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for i in range(iters):
                ops[i % NW if rot_w else 0](xs[i % NX if rot_x else 0], out=out)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print(f"conv3x3 {C}->{C} @{S} B={B}, graph of 320 launches, us per launch:")
print(f"  same weights, same input     {run(False, False):6.2f}")
print(f"  rotating weights (16 sets)   {run(True, False):6.2f}")
print(f"  rotating inputs (16 tensors) {run(False, True):6.2f}")
print(f"  both                         {run(True, True):6.2f}")
