"""The allocation-free loss entry point (ldm_mse_noalloc) against the entry point it is a variant of (ldm_mse),
through the C ABI: same bits, two calls in a row, the second on a smaller input."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mse(lib, fn_noalloc, a, b):
    from diffusionmodel_amd import _lib
    out = torch.full((1,), -1.0, dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    if fn_noalloc:
        _lib.check(lib.ldm_mse_noalloc(p(a), p(b), a.numel(), p(out), st))
    else:
        ws = torch.empty(1024, dtype=torch.float32, device="cuda")
        _lib.check(lib.ldm_mse(p(a), p(b), a.numel(), p(ws), p(out), st))
    torch.cuda.synchronize()
    return out


def test_mse_noalloc_matches_mse_on_two_calls_in_a_row():
    from diffusionmodel_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(12)
    # the second input is smaller: fewer elements per partial sum, and with 1000 elements most of the 1024 partial
    # sums are sums over nothing; what the first call left in the array must not show
    for n in (64 * 64 * 64 * 4 + 3, 1000):
        a = torch.randn(n, generator=g).cuda()
        b = torch.randn(n, generator=g).cuda()
        print(f"mse_noalloc n={n}", flush=True)
        ref = _mse(lib, False, a, b)
        got = _mse(lib, True, a, b)
        assert torch.equal(got, ref), (n, got.item(), ref.item())
        assert abs(ref.item() - ((a - b) ** 2).mean().item()) < 1e-4 * ref.item()
    # beyond the stated maximum: rejected on the host, nothing launched
    a = torch.zeros(8, device="cuda")
    out = torch.zeros(1, device="cuda")
    rc = lib.ldm_mse_noalloc(C.c_void_p(a.data_ptr()), C.c_void_p(a.data_ptr()), (1 << 36) + 1, C.c_void_p(out.data_ptr()),
                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 1 and b"LDM_MSE_NOALLOC_MAX_N" in lib.ldm_last_error()


def test_mse_noalloc_allocates_nothing_and_replays_from_a_graph():
    """With the output preallocated, device memory in use is the same before and after the call; after the set-up
    call (ldm_mse_noalloc_prepare) the entry point can be captured into a graph, and every replay -- with the
    program's own eager work and synchronisations in between -- gives the bits of the eager call."""
    from diffusionmodel_amd import _lib
    lib = _lib.load()
    _lib.check(lib.ldm_mse_noalloc_prepare())
    g = torch.Generator().manual_seed(13)
    n = 64 * 64 * 64 + 5
    a = torch.randn(n, generator=g).cuda()
    b = torch.randn(n, generator=g).cuda()
    out = torch.full((1,), -1.0, dtype=torch.float32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    alloc0 = torch.cuda.memory_allocated()
    _lib.check(lib.ldm_mse_noalloc(p(a), p(b), n, p(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0 and torch.cuda.memory_allocated() == alloc0
    eager = out.clone()
    assert torch.equal(eager, _mse(lib, False, a, b))
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    out.fill_(-1.0)
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            _lib.check(lib.ldm_mse_noalloc(p(a), p(b), n, p(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for k in range(3):
        out.fill_(-1.0)
        other = _mse(lib, False, b, a * float(k + 2))          # eager work of the program between the replays
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), (k, out.item(), eager.item())
        assert other.item() > 0
