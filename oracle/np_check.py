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


def batch_norm_train(x, gamma, beta, eps=1e-5):
    """Training-mode BatchNorm by explicit loops: per channel the mean and the biased variance over every (image,
    row, column), then ((x - mean) / sqrt(var + eps)) * gamma + beta.  Returns (y, mean, var)."""
    x = np.asarray(x, np.float64)
    B, C, H, W = x.shape
    y = np.zeros_like(x)
    mean = np.zeros(C); var = np.zeros(C)
    n = B * H * W
    for c in range(C):
        acc = 0.0
        for b in range(B):
            for i in range(H):
                for j in range(W):
                    acc += x[b, c, i, j]
        mean[c] = acc / n
        acc = 0.0
        for b in range(B):
            for i in range(H):
                for j in range(W):
                    acc += (x[b, c, i, j] - mean[c]) ** 2
        var[c] = acc / n
        y[:, c] = (x[:, c] - mean[c]) / np.sqrt(var[c] + eps) * float(gamma[c]) + float(beta[c])
    return y, mean, var


def philox4x32_10_scalar(ctr, key):
    """One Philox-4x32-10 block with Python integers (Salmon et al. 2011): ctr 4 words, key 2 words -> 4 words."""
    c0, c1, c2, c3 = (int(v) & 0xFFFFFFFF for v in ctr)
    k0, k1 = (int(v) & 0xFFFFFFFF for v in key)
    for _ in range(10):
        p0 = c0 * 0xD2511F53
        p1 = c2 * 0xCD9E8D57
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def dropout_keep(seed, b, c, y, x, C, W, keep_prob):
    """Whether element (b, c, y, x) survives the build's dropout (see ldm_oracle.dropout_mask_nchw)."""
    e = (y * W + x) * C + c
    w = philox4x32_10_scalar((e // 4, 0xFFFFFFFE, b, 0x4C444D32), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[e % 4]
    u = np.float32(np.float32((w >> 8) + 0.5) * np.float32(2.0 ** -24))
    return bool(u < np.float32(keep_prob))


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
