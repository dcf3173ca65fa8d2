"""Kernel-level operator: one convolution of the UNet on NHWC tensors in the library's own layout
(ldm_conv_* in include/ldm_hip.h).  Used for per-kernel parity tests and for roofline timing."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

KINDS = {"conv3x3": 0, "conv3x3_s2": 1, "convT4x4_s2": 2, "conv1x1": 3}
ACTS = {None: 0, "none": 0, "silu": 1, "relu": 2}


def to_nhwc(x: torch.Tensor, dtype: str) -> torch.Tensor:
    """NCHW f32 -> contiguous NHWC tensor of the operator's element type."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    return x.permute(0, 2, 3, 1).contiguous().to(td)


def from_nhwc(y: torch.Tensor) -> torch.Tensor:
    return y.float().permute(0, 3, 1, 2).contiguous()


class ConvOp:
    """kind: 'conv3x3' | 'conv3x3_s2' | 'convT4x4_s2' | 'conv1x1'; weight in burn's layout for the kind
    ([Cout,Cin,3,3] | [Cin,Cout,4,4] | [Cout,Cin0+Cin1(,1,1)]); channel counts multiples of 64."""

    def __init__(self, kind: str, weight: torch.Tensor, bias: torch.Tensor, dtype: str = "bf16", cin1: int = 0):
        self._lib = _lib.load()
        self.kind, self.dtype = kind, dtype
        self.code = _lib.LDM_BF16 if dtype == "bf16" else _lib.LDM_F32
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous()
        if kind == "convT4x4_s2":
            cin, self.cout = w.shape[0], w.shape[1]
        else:
            self.cout, cin = w.shape[0], w.shape[1]
        self.cin0, self.cin1 = cin - cin1, cin1
        self._h = C.c_void_p()
        _lib.check(self._lib.ldm_conv_create(C.byref(self._h), KINDS[kind], self.code, self.cin0, self.cin1, self.cout,
                                             C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr())))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.ldm_conv_destroy(h)
            except Exception:
                pass

    def out_hw(self, H: int, W: int):
        if self.kind == "conv3x3_s2":
            return (H - 1) // 2 + 1, (W - 1) // 2 + 1
        if self.kind == "convT4x4_s2":
            return 2 * H, 2 * W
        return H, W

    def forward(self, x0: torch.Tensor, x1: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                act: Optional[str] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x0 [B,H,W,cin0], x1 [B,H,W,cin1] or None, res [B,Ho,Wo,cout] or None: contiguous NHWC device tensors."""
        td = torch.bfloat16 if self.dtype == "bf16" else torch.float32
        for t, c in ((x0, self.cin0), (x1, self.cin1)):
            if t is None:
                continue
            if not (t.is_cuda and t.is_contiguous() and t.dtype == td and t.dim() == 4 and t.shape[3] == c):
                raise ValueError("inputs must be contiguous NHWC device tensors of the operator's dtype and channels")
        if (x1 is None) != (self.cin1 == 0):
            raise ValueError("second source does not match the operator")
        B, H, W, _ = x0.shape
        Ho, Wo = self.out_hw(H, W)
        if out is None:
            out = torch.empty((B, Ho, Wo, self.cout), dtype=td, device=x0.device)
        if res is not None and not (res.is_contiguous() and res.dtype == td and tuple(res.shape) == tuple(out.shape)):
            raise ValueError("residual must match the output")
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
        _lib.check(self._lib.ldm_conv_forward(self._h, p(x0), p(x1), p(res), p(out), B, H, W, ACTS[act], stream))
        return out

    __call__ = forward


def batchnorm_act(h: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, act: Optional[str] = None,
                  res: Optional[torch.Tensor] = None, tvec: Optional[torch.Tensor] = None, eps: float = 1e-5,
                  dropout_p: float = 0.0, seed: int = 0, out: Optional[torch.Tensor] = None, stats_only: bool = False):
    """Training-mode BatchNorm + the ResidualBlock's tail on a contiguous NHWC device tensor h [B,H,W,C] (bf16 or
    f32): out = dropout(act(((h - mean) / sqrt(var + eps)) * gamma + beta + res) + tvec[b]) with the batch
    statistics of h (ldm_batchnorm_act; burn's BatchNorm on the autodiff backend, src/main.rs:374, 385, 1318).
    Returns (out, mean, var); out is None with stats_only."""
    if not (h.is_cuda and h.is_contiguous() and h.dim() == 4 and h.dtype in (torch.bfloat16, torch.float32)):
        raise ValueError("h must be a contiguous NHWC device tensor, bf16 or f32")
    B, H, W, Cc = h.shape
    code = _lib.LDM_BF16 if h.dtype == torch.bfloat16 else _lib.LDM_F32
    lib = _lib.load()
    g = gamma.detach().to(h.device, torch.float32).contiguous()
    b = beta.detach().to(h.device, torch.float32).contiguous()
    if g.numel() != Cc or b.numel() != Cc:
        raise ValueError("gamma and beta must hold one value per channel")
    nws = lib.ldm_batchnorm_ws_floats(code, B * H * W, Cc)
    if nws <= 0:
        raise ValueError("channel count not served")
    ws = torch.empty(nws, dtype=torch.float32, device=h.device)
    mean = torch.empty(Cc, dtype=torch.float32, device=h.device)
    var = torch.empty(Cc, dtype=torch.float32, device=h.device)
    if res is not None and not (res.is_contiguous() and res.dtype == h.dtype and tuple(res.shape) == tuple(h.shape)):
        raise ValueError("res must match h")
    if tvec is not None:
        tvec = tvec.detach().to(h.device, torch.float32).contiguous()
        if tuple(tvec.shape) != (B, Cc):
            raise ValueError("tvec must be [B, C]")
    if not stats_only and out is None:
        out = torch.empty_like(h)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
    _lib.check(lib.ldm_batchnorm_act(code, p(h), p(res), p(None if stats_only else out), B, H * W, Cc, p(g), p(b), eps,
                                     ACTS[act], p(tvec), dropout_p, seed & (2 ** 64 - 1), p(ws), p(mean), p(var), stream))
    return (None if stats_only else out), mean, var
