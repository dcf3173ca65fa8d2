"""Dependent chain of one convolution shape (x -> y -> x -> ...): per-launch time including what a launch leaves
for the next one (dirty lines, cache residency).  usage: chainbench.py kind B S C [iters]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffusionmodel_amd.ops import ConvOp
kind, B, S, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
g = torch.Generator().manual_seed(0)
ops = [ConvOp(kind, torch.randn(C, C, 3, 3, generator=g) * (1.0 / (3 * C ** 0.5)), torch.randn(C, generator=g) * 0.05) for _ in range(2)]
x = torch.randn(B, S, S, C, device="cuda").bfloat16()
y = torch.empty_like(x)
def run(n):
    a, b = x, y
    for i in range(n):
        ops[i & 1](a, out=b)
        a, b = b, a
run(10); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(iters); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print(f"{os.environ.get('LDM_HIP_LIB','default')[-20:]:>20} chain {kind} B{B} S{S} {C}->{C}: {us:7.2f} us per launch  {2.0*B*S*S*C*C*9/us/1e6:7.1f} TF/s", flush=True)
