import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diffusionmodel_amd.ops import ConvOp
def bench(kind, B, S, Cin, Cout, dtype="bf16", res=False, iters=50, act=None):
    g = torch.Generator().manual_seed(0)
    if kind == "convT4x4_s2":
        w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.05
    elif kind == "conv1x1":
        w = torch.randn(Cout, Cin, generator=g) * 0.05
    else:
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.05
    op = ConvOp(kind, w, b, dtype=dtype)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    x = torch.randn(B, S, S, Cin, device="cuda").to(td)
    Ho, Wo = op.out_hw(S, S)
    out = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=td)
    r = torch.randn(B, Ho, Wo, Cout, device="cuda").to(td) if res else None
    for _ in range(5): op(x, res=r, out=out, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): op(x, res=r, out=out, act=act)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    taps = {"conv3x3": 9, "conv3x3_s2": 9, "convT4x4_s2": 4, "conv1x1": 1}[kind]
    flop = 2.0 * B * Ho * Wo * Cin * Cout * taps
    print(f"{os.environ.get('LDM_HIP_LIB','default')[-24:]:>24} {kind:12s} B{B} S{S} {Cin}->{Cout} res={int(res)} act={act} {dtype}: {ms*1e3:8.1f} us  {flop/ms/1e9:8.1f} TF/s", flush=True)
if __name__ == "__main__":
    B = int(os.environ.get("CB", "64"))
    bench("conv3x3", B, 64, 64, 64)
    bench("conv3x3", B, 64, 64, 64, res=True)
    bench("conv3x3", B, 64, 64, 64, res=True, act="silu")
    bench("conv3x3", B, 32, 128, 128, res=True, act="silu")
    bench("conv3x3", B, 16, 256, 256, res=True, act="silu")
    bench("conv3x3", B, 32, 128, 128)
    bench("conv3x3", B, 16, 256, 256)
    bench("conv3x3_s2", B, 64, 64, 128)
    bench("conv3x3_s2", B, 32, 128, 256)
    bench("convT4x4_s2", B, 16, 256, 128)
    bench("convT4x4_s2", B, 32, 128, 64)
    bench("conv1x1", B, 32, 256, 128)
    bench("conv1x1", B, 64, 128, 64)
    bench("conv1x1", B, 16, 256, 768)
