"""Second, independent statement of the operators used by oracle/ldm_oracle.py.  TEST INFRASTRUCTURE ONLY.

Direct-loop numpy/float64 code written from the operator definitions (cross-correlation with zero
padding; transposed convolution out = (in-1)*s - 2p + k; soft-max attention), not from PyTorch.
It exists so that the PyTorch restatement is not checked only against itself.  Tiny shapes only.
"""
from __future__ import annotations

import numpy as np


def conv2d(x, w, b, stride, pad):
    """x [B,Ci,H,W], w [Co,Ci,kh,kw] -> [B,Co,Ho,Wo]; y[o,i,j] = b[o] + sum x[c, i*s-p+u, j*s-p+v] w[o,c,u,v]."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64); b = np.asarray(b, np.float64)
    B, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    y = np.zeros((B, Co, Ho, Wo))
    for n in range(B):
        for o in range(Co):
            for i in range(Ho):
                for j in range(Wo):
                    acc = b[o]
                    for u in range(kh):
                        yy = i * stride - pad + u
                        if yy < 0 or yy >= H:
                            continue
                        for v in range(kw):
                            xx = j * stride - pad + v
                            if xx < 0 or xx >= W:
                                continue
                            acc += np.dot(x[n, :, yy, xx], w[o, :, u, v])
                    y[n, o, i, j] = acc
    return y


def conv_transpose2d(x, w, b, stride, pad):
    """x [B,Ci,H,W], w [Ci,Co,kh,kw]: every input pixel scatters x*w into out[i*s-p+u, j*s-p+v]."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64); b = np.asarray(b, np.float64)
    B, Ci, H, W = x.shape
    _, Co, kh, kw = w.shape
    Ho = (H - 1) * stride - 2 * pad + kh
    Wo = (W - 1) * stride - 2 * pad + kw
    y = np.zeros((B, Co, Ho, Wo))
    y += b.reshape(1, Co, 1, 1)
    for n in range(B):
        for i in range(H):
            for j in range(W):
                for u in range(kh):
                    oy = i * stride - pad + u
                    if oy < 0 or oy >= Ho:
                        continue
                    for v in range(kw):
                        ox = j * stride - pad + v
                        if ox < 0 or ox >= Wo:
                            continue
                        y[n, :, oy, ox] += x[n, :, i, j] @ w[:, :, u, v]
    return y


def batch_norm(x, gamma, beta, mean, var, eps=1e-5):
    x = np.asarray(x, np.float64)
    sh = (1, -1, 1, 1)
    return (x - np.asarray(mean, np.float64).reshape(sh)) / np.sqrt(np.asarray(var, np.float64).reshape(sh) + eps) \
        * np.asarray(gamma, np.float64).reshape(sh) + np.asarray(beta, np.float64).reshape(sh)


def silu(x):
    return x / (1.0 + np.exp(-x))


def attention(x, wq, bq, wk, bk, wv, bv, wo, bo):
    """x [B,C,H,W]; weights [C_in, C_out].  Token n = y*W + x; softmax over keys; residual add."""
    x = np.asarray(x, np.float64)
    B, C, H, W = x.shape
    out = np.zeros_like(x)
    for n in range(B):
        tok = x[n].reshape(C, H * W).T                     # [N, C]
        q = tok @ np.asarray(wq, np.float64) + np.asarray(bq, np.float64)
        k = tok @ np.asarray(wk, np.float64) + np.asarray(bk, np.float64)
        v = tok @ np.asarray(wv, np.float64) + np.asarray(bv, np.float64)
        o = np.zeros_like(q)
        for i in range(H * W):
            s = np.array([np.dot(q[i], k[j]) for j in range(H * W)]) / np.sqrt(float(C))
            e = np.exp(s - s.max())
            pr = e / e.sum()
            o[i] = pr @ v
        o = o @ np.asarray(wo, np.float64) + np.asarray(bo, np.float64)
        out[n] = o.T.reshape(C, H, W) + x[n]
    return out
