"""Diagnostic: torch's scaled_dot_product_attention (whatever backend it picks on this build) on the single-head
shapes of the attention blocks, for comparison with attention.hip."""
import torch, torch.nn.functional as F
def t(B, N, D, iters=10):
    q, k, v = (torch.randn(B, 1, N, D, device="cuda").bfloat16() for _ in range(3))
    try:
        for _ in range(2): F.scaled_dot_product_attention(q, k, v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): F.scaled_dot_product_attention(q, k, v)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"sdpa B={B} N={N} D={D} bf16: {ms*1e3:9.1f} us  {4.0*B*N*N*D/ms/1e9:7.1f} TF/s", flush=True)
    except Exception as e:
        print(f"sdpa B={B} N={N} D={D}: {type(e).__name__}: {str(e)[:120]}", flush=True)
t(64, 256, 256, 50); t(16, 16384, 256, 3); t(16, 4096, 512, 5)
