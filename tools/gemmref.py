"""Diagnostic: the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) on the im2col shapes of the three 3x3 convolutions
of the benchmark, for comparison with the implicit-GEMM kernels (the library does not have to gather a window)."""
import torch
def t(M, K, N, iters=50):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16()
    for _ in range(5): torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): torch.mm(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"mm [{M},{K}] x [{K},{N}] bf16: {ms*1e3:7.1f} us  {2.0*M*K*N/ms/1e9:7.1f} TF/s", flush=True)
t(262144, 576, 64); t(65536, 1152, 128); t(16384, 2304, 256); t(8192, 8192, 8192, 10)
