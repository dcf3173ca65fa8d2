"""CPU restatement of the reference's sampling hot path.  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it.  The shipped package (diffusionmodel_amd) never does.

PARITY UNPINNED: the reference (ragu812/diffusionModel) holds no tests, golden vectors or fixtures
for this path, its Python file does not parse and its Rust path cannot be built in this image, so
nothing below could be checked against outputs of the reference itself.  What pins this file instead:
  * every function follows the cited line range of /root/reference/src/main.rs;
  * an independent direct-loop numpy/float64 statement of the same operators (oracle/np_check.py)
    agrees with it (tests/test_oracle.py);
  * the schedule known-answer values that can be derived by hand from src/main.rs:553-583.

Decisions that the reference leaves open (its Unet::forward cannot run as written):
  D1  down-sampling is done by the stride-2 convolutions down2/down3; the 2x2 max-pools pool1/pool2
      (src/main.rs:432,443,502,506) are skipped, which is the only pyramid consistent with the two
      x2 transposed convolutions and both skip concatenations (src/main.rs:515-516,520-521).
  D2  attention follows the Rust statement (softmax over keys, out-projection applied).
  D3  the SAMPLER has inference semantics: BatchNorm uses running statistics, Dropout is the identity.  The training
      forward (DiffusionModel::forward, src/main.rs:635-679, only ever run on the autodiff backend, :1318) is stated
      with `train=True`: batch-statistics BatchNorm and active Dropout, following burn 0.18.0's published
      BatchNorm::forward_train / Dropout::forward (burn-core nn/norm/batch.rs, nn/dropout.rs; a Cargo.lock dependency
      whose source is not in /root/reference, so this part is restated from the published algorithm).
  D4  BatchNorm eps = 1e-5 (burn default).
  D5  the per-step NaN probes (src/main.rs:706-710,725-729,759-763) are not reproduced.
  D7  "scaled" UNet = same topology with wider channels plus one extra Attention after up_res1.

Weight layouts (burn's): Conv2d [Cout,Cin,kh,kw]; ConvTranspose2d [Cin,Cout,kh,kw]; Linear [d_in,d_out].
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------------
# a1  sinusoidal embedding                                           src/main.rs:309-335
# --------------------------------------------------------------------------------------------
def sin_time_addition(time: torch.Tensor, dim: int, dtype=torch.float32) -> torch.Tensor:
    """emb[b, i] = sin(t_b * f_i), emb[b, half+i] = cos(t_b * f_i), f_i = exp(-(i*4*ln10/half)).

    The frequencies are built on the host in f32 exactly as src/main.rs:318-323 does
    (f32 multiply, f32 divide, f32 exp); the product and sin/cos are in `dtype`.
    """
    half = dim // 2
    i = np.arange(half, dtype=np.float32)
    expo = i * np.float32(4.0) * np.float32(math.log(10.0)) / np.float32(half)
    freqs = np.exp(-expo).astype(np.float32)
    f = torch.from_numpy(freqs).to(dtype)
    arg = time.to(dtype).reshape(-1, 1) * f.reshape(1, half)
    return torch.cat([arg.sin(), arg.cos()], dim=1)


# --------------------------------------------------------------------------------------------
# a2  time MLP                                                       src/main.rs:278-307
# --------------------------------------------------------------------------------------------
def linear(p: Dict[str, torch.Tensor], name: str, x: torch.Tensor) -> torch.Tensor:
    """burn Linear: y = x @ W + b with W stored [d_in, d_out]."""
    return x @ p[name + ".weight"].to(x.dtype) + p[name + ".bias"].to(x.dtype)


def time_addition(p, prefix: str, t: torch.Tensor) -> torch.Tensor:
    z = linear(p, prefix + ".linear1", t)
    z = F.relu(z)
    return linear(p, prefix + ".linear2", z)


# --------------------------------------------------------------------------------------------
# helpers: conv / batch-norm (third-party burn semantics: cross-correlation, NCHW, zero pad)
# --------------------------------------------------------------------------------------------
def conv2d(p, name: str, x, stride: int, padding: int):
    return F.conv2d(x, p[name + ".weight"].to(x.dtype), p[name + ".bias"].to(x.dtype),
                    stride=stride, padding=padding)


def conv_transpose2d(p, name: str, x, stride: int, padding: int):
    return F.conv_transpose2d(x, p[name + ".weight"].to(x.dtype), p[name + ".bias"].to(x.dtype),
                              stride=stride, padding=padding)


def batch_norm(p, name: str, x):
    """Inference BatchNorm (D3/D4): (x - mean) / sqrt(var + eps) * gamma + beta."""
    g = p[name + ".gamma"].to(x.dtype).reshape(1, -1, 1, 1)
    b = p[name + ".beta"].to(x.dtype).reshape(1, -1, 1, 1)
    m = p[name + ".running_mean"].to(x.dtype).reshape(1, -1, 1, 1)
    v = p[name + ".running_var"].to(x.dtype).reshape(1, -1, 1, 1)
    return (x - m) / torch.sqrt(v + BN_EPS) * g + b


def batch_norm_train(p, name: str, x, stats: Optional[dict] = None):
    """burn 0.18 BatchNorm::forward_train (the branch taken when the backend has autodiff enabled): per channel
    mean over (B, H, W), var = mean of (x - mean)^2 over the same elements (biased), then forward_shared:
    ((x - mean) / sqrt(var + eps)) * gamma + beta.  The running statistics are updated as a side effect there
    (running = running * (1 - 0.1) + batch * 0.1); here the batch statistics are recorded in `stats[name]`."""
    mean = x.mean(dim=(0, 2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    if stats is not None:
        stats[name] = (mean.reshape(-1).clone(), var.reshape(-1).clone())
    g = p[name + ".gamma"].to(x.dtype).reshape(1, -1, 1, 1)
    b = p[name + ".beta"].to(x.dtype).reshape(1, -1, 1, 1)
    return (x - mean) / torch.sqrt(var + BN_EPS) * g + b


DROPOUT_TAG = 0x4C444D32
DROPOUT_STEP = 0xFFFFFFFE


def dropout_mask_nchw(seed: int, B: int, C: int, H: int, W: int, keep_prob: float) -> torch.Tensor:
    """The keep mask of the build's dropout, [B, C, H, W] of 0 / 1 (f32).  burn's Dropout::forward draws
    Bernoulli(1 - p) from the backend's RNG (unreproducible); the build draws it from the counter-based generator so
    that both sides can hold the same mask: element (b, c, y, x) is kept when word (c % 4) of Philox-4x32-10 with
    counter {((y W + x) C + c) // 4, DROPOUT_STEP, b, DROPOUT_TAG} and key (seed low, seed high) gives
    ((word >> 8) + 0.5) * 2^-24 < keep_prob."""
    assert C % 4 == 0
    G = H * W * C // 4
    g = np.arange(G, dtype=np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    out = np.empty((B, H, W, C), dtype=np.float32)
    for b in range(B):
        ctr = np.stack([g, np.full(G, DROPOUT_STEP, np.uint32), np.full(G, b, np.uint32),
                        np.full(G, DROPOUT_TAG, np.uint32)], axis=-1)
        r = philox4x32_10(ctr, key)
        u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
        out[b] = (u < np.float32(keep_prob)).astype(np.float32).reshape(H, W, C)
    return torch.from_numpy(out).permute(0, 3, 1, 2).contiguous()


def dropout(x, prob: float, seed: int):
    """burn 0.18 Dropout::forward with autodiff enabled: x * Bernoulli(1 - prob) * (1 / (1 - prob))."""
    if prob == 0.0:
        return x
    keep = 1.0 - prob
    mask = dropout_mask_nchw(seed, x.shape[0], x.shape[1], x.shape[2], x.shape[3], keep).to(x.dtype)
    return x * mask * (1.0 / keep)


# --------------------------------------------------------------------------------------------
# a3-a5  ResidualBlock                                               src/main.rs:372-387
# --------------------------------------------------------------------------------------------
def residual_block(p, prefix: str, x, t_emb, train: bool = False, stats: Optional[dict] = None):
    bn = (lambda name, h: batch_norm_train(p, name, h, stats)) if train else (lambda name, h: batch_norm(p, name, h))
    h = conv2d(p, prefix + ".conv1", x, 1, 1)
    h = bn(prefix + ".bn1", h)
    h = F.silu(h)
    t = linear(p, prefix + ".time", t_emb)
    t = F.silu(t)
    h = h + t.reshape(t.shape[0], t.shape[1], 1, 1)
    h = conv2d(p, prefix + ".conv2", h, 1, 1)
    h = bn(prefix + ".bn2", h)
    return F.silu(h + x)


# --------------------------------------------------------------------------------------------
# a8  Attention                                                      src/main.rs:256-276
# --------------------------------------------------------------------------------------------
def attention(p, prefix: str, x, q_chunk: int = 2048):
    """q_chunk only bounds host memory: the [tokens, tokens] score matrix is built `q_chunk` query rows at a
    time.  Every row still sees all keys, so each soft-max is the exact one of src/main.rs:267-269."""
    a, b, c, d = x.shape
    xf = x.reshape(a, b, c * d).transpose(1, 2)          # [B, tokens, C]
    q = linear(p, prefix + ".question", xf)
    k = linear(p, prefix + ".key", xf)
    v = linear(p, prefix + ".val", xf)
    scaling = math.sqrt(float(b))                        # sqrt(channels)
    n = c * d
    if n <= q_chunk:
        score = q @ k.transpose(1, 2) / scaling
        w = torch.softmax(score, dim=2)                  # over keys
        o = w @ v
    else:
        o = torch.empty_like(q)
        kt = k.transpose(1, 2)
        for q0 in range(0, n, q_chunk):
            score = q[:, q0:q0 + q_chunk] @ kt / scaling
            o[:, q0:q0 + q_chunk] = torch.softmax(score, dim=2) @ v
    o = linear(p, prefix + ".out", o)
    o = o.transpose(1, 2).reshape(a, b, c, d)
    return o + x


# --------------------------------------------------------------------------------------------
# SimpleDecoder::forward (= Vae::decode)                              src/main.rs:149-169, 208-210
# --------------------------------------------------------------------------------------------
def vae_decode(p, prefix: str, z):
    """z [B, latent_dimen, 4, 4] -> [B, output, 64, 64] in [-1, 1].  Dropout (:154) is the identity (D3),
    BatchNorm uses running statistics (D3/D4)."""
    pre = prefix + "." if prefix else ""
    y = z.flatten(1, 3)                                   # :150
    y = F.relu(linear(p, pre + "transfer", y))            # :151-152
    y = y.reshape(y.shape[0], 128, 16, 16)                # :156-157
    y = F.relu(batch_norm(p, pre + "bn_1", conv_transpose2d(p, pre + "reverse1", y, 2, 1)))   # :159-161
    y = F.relu(batch_norm(p, pre + "bn_2", conv_transpose2d(p, pre + "reverse2", y, 2, 1)))   # :163-165
    y = conv_transpose2d(p, pre + "reverse3", y, 1, 1)    # :167
    return torch.tanh(y)                                  # :168


def to_rgb8(img) -> np.ndarray:
    """save_as_image's pixel arithmetic (src/main.rs:1297-1308) for every image: [B,C,H,W] f32 ->
    [B,H,W,C] u8, ((v + 1.0) * 127.5).clamp(0, 255) as u8 in f32 (Rust's `as u8` truncates; NaN -> 0)."""
    v = np.asarray(img, dtype=np.float32)
    v = (v + np.float32(1.0)) * np.float32(127.5)
    v = np.where(np.isnan(v), np.float32(0.0), np.clip(v, np.float32(0.0), np.float32(255.0)))
    return np.ascontiguousarray(v.astype(np.uint8).transpose(0, 2, 3, 1))


# --------------------------------------------------------------------------------------------
# Unet::forward                                                      src/main.rs:496-526 (+D1, D7)
# --------------------------------------------------------------------------------------------
def unet_forward(p, x, time, scaled_attention: bool = False, return_time_embed: bool = False, train: bool = False,
                 dropout_p: float = 0.5, dropout_seed: int = 0, stats: Optional[dict] = None):
    """train=True: the forward of the training step (autodiff backend, src/main.rs:1318): BatchNorm on batch
    statistics, mid_drop (:458, :513) active; `stats` collects {bn name: (batch mean, batch var)}."""
    rb = lambda name, h: residual_block(p, name, h, te, train=train, stats=stats)   # noqa: E731
    te = time_addition(p, "time_emb", time)
    d1 = conv2d(p, "down1", x, 1, 1)
    d1 = rb("down_res1", d1)
    d2 = conv2d(p, "down2", d1, 2, 1)                    # pool1 skipped (D1)
    d2 = rb("down_res2", d2)
    d3 = conv2d(p, "down3", d2, 2, 1)                    # pool2 skipped (D1)
    d3 = rb("down_res3", d3)
    m = rb("mid_res1", d3)
    m = attention(p, "mid_atten", m)
    m = rb("mid_res2", m)
    if train:
        m = dropout(m, dropout_p, dropout_seed)          # mid_drop, :513; identity at inference (D3)
    u1 = conv_transpose2d(p, "up1", m, 2, 1)
    u1 = torch.cat([u1, d2], dim=1)
    u1 = conv2d(p, "up_conv1", u1, 1, 0)
    u1 = rb("up_res1", u1)
    if scaled_attention:
        u1 = attention(p, "up_atten", u1)                # D7
    u2 = conv_transpose2d(p, "up2", u1, 2, 1)
    u2 = torch.cat([u2, d1], dim=1)
    u2 = conv2d(p, "up_conv2", u2, 1, 0)
    u2 = rb("up_res2", u2)
    out = conv2d(p, "out_conv", u2, 1, 1)
    if return_time_embed:
        return out, te
    return out


# --------------------------------------------------------------------------------------------
# schedule                                                           src/main.rs:553-583
# --------------------------------------------------------------------------------------------
def get_betas(num_time: int) -> np.ndarray:
    """Linear 1e-4 -> 0.02, clamped to [1e-6, 0.999]; every operation in f32 as at :558-564."""
    b0 = np.float32(1e-4)
    b1 = np.float32(0.02)
    den = np.float32(max(num_time - 1, 1))
    out = np.empty(num_time, dtype=np.float32)
    for i in range(num_time):
        ratio = np.float32(i) / den
        beta = np.float32(b0 + np.float32(b1 - b0) * ratio)
        out[i] = min(max(beta, np.float32(1e-6)), np.float32(0.999))
    return out


def get_alphas(num_time: int) -> np.ndarray:
    """Returns alpha-BAR (the running product), as the reference's get_alphas does (:569-583)."""
    betas = get_betas(num_time)
    alphas = (np.float32(1.0) - betas).astype(np.float32)
    out = np.empty(num_time, dtype=np.float32)
    c = np.float32(1.0)
    for i in range(num_time):
        c = np.float32(c * alphas[i])
        out[i] = c
    return out


def ddpm_coefficients(num_time: int) -> np.ndarray:
    """Per-timestep scalars of the reverse step, f32, operation order of src/main.rs:712-747.

    Row i = [sqrt(abar_i), sqrt(1-abar_i), coef1, coef2, sigma]; for i == 0 the last three are
    (1, 0, 0) and the caller uses z = x0 (:765).
    """
    f = np.float32
    betas, abar = get_betas(num_time), get_alphas(num_time)
    tab = np.zeros((num_time, 5), dtype=np.float32)
    for i in range(num_time):
        a_t, b_t = abar[i], betas[i]
        tab[i, 0] = np.sqrt(a_t)
        tab[i, 1] = np.sqrt(f(a_t * f(-1.0)) + f(1.0))
        if i > 0:
            a_p = abar[i - 1]
            om_t = f(f(a_t * f(-1.0)) + f(1.0))
            om_p = f(f(a_p * f(-1.0)) + f(1.0))
            var = f(f(om_p / f(om_t + f(1e-8))) * b_t)
            tab[i, 4] = np.sqrt(var)
            tab[i, 2] = f(f(np.sqrt(a_p) * b_t) / f(om_t + f(1e-8)))
            tab[i, 3] = f(f(np.sqrt(f(f(1.0) - b_t)) * om_p) / f(om_t + f(1e-8)))
        else:
            tab[i, 2], tab[i, 3], tab[i, 4] = 1.0, 0.0, 0.0
    return tab


def ddpm_step(z, eps, i: int, coef: np.ndarray, noise: Optional[torch.Tensor]):
    """One reverse step (src/main.rs:712-756,765).  `coef` = ddpm_coefficients(num_time)."""
    sa, s1m, c1, c2, sig = (float(v) for v in coef[i])
    x0 = (z - eps * s1m) / sa
    if i > 0:
        mean = x0 * c1 + z * c2
        return mean + sig * noise
    return x0


# --------------------------------------------------------------------------------------------
# a11  DDIM (eta = 0) -- NOT in the reference; build-defined, parity unpinned by construction
# --------------------------------------------------------------------------------------------
def ddim_timesteps(num_time: int, steps: int) -> List[int]:
    """`steps` timesteps evenly strided over [0, num_time): t_k = k * (num_time // steps),
    visited from k = steps-1 down to 0 ("leading" spacing)."""
    stride = num_time // steps
    assert stride >= 1
    return [k * stride for k in range(steps - 1, -1, -1)]


def ddim_coefficients(num_time: int, steps: int) -> np.ndarray:
    """Row k (execution order) = [sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev), 0]
    with abar_prev = abar[t - stride], and 1 when t - stride < 0 (so the last step returns x0)."""
    f = np.float32
    abar = get_alphas(num_time)
    ts = ddim_timesteps(num_time, steps)
    stride = num_time // steps
    tab = np.zeros((steps, 5), dtype=np.float32)
    for k, t in enumerate(ts):
        a_t = abar[t]
        a_p = abar[t - stride] if t - stride >= 0 else f(1.0)
        tab[k, 0] = np.sqrt(a_t)
        tab[k, 1] = np.sqrt(f(f(a_t * f(-1.0)) + f(1.0)))
        tab[k, 2] = np.sqrt(a_p)
        tab[k, 3] = np.sqrt(f(f(a_p * f(-1.0)) + f(1.0)))
    return tab


def ddim_step(z, eps, k: int, coef: np.ndarray):
    sa, s1m, sap, s1mp, _ = (float(v) for v in coef[k])
    x0 = (z - eps * s1m) / sa
    return x0 * sap + eps * s1mp


# --------------------------------------------------------------------------------------------
# denoising_process                                                  src/main.rs:681-777
# --------------------------------------------------------------------------------------------
def denoising_process(p, z0: torch.Tensor, num_time: int, noise: Optional[torch.Tensor] = None,
                      mode: str = "ddpm", ddim_steps: int = 50, time_dim: int = 256,
                      scaled_attention: bool = False, trace: bool = False,
                      z_override: Optional[Sequence[torch.Tensor]] = None):
    """Runs the sampling loop on the given initial latent.

    noise: [steps, B, C, H, W]; row k is used at the k-th executed step (DDPM only; the last
           step, i == 0, uses none).  mode "ddim" is the build-defined eta=0 sampler.
    trace: also return the per-step (z_in, eps) lists.
    The trailing vae.decode (src/main.rs:786) is outside the path.
    """
    z = z0.clone()
    dt = z.dtype
    if mode == "ddpm":
        ts = list(range(num_time - 1, -1, -1))
        coef = ddpm_coefficients(num_time)
    else:
        ts = ddim_timesteps(num_time, ddim_steps)
        coef = ddim_coefficients(num_time, ddim_steps)
    zs, epss = [], []
    B = z.shape[0]
    for k, t in enumerate(ts):
        if z_override is not None:
            z = z_override[k].clone()
        temb = sin_time_addition(torch.full((B,), t, dtype=torch.int64), time_dim, dtype=dt)
        eps = unet_forward(p, z, temb, scaled_attention=scaled_attention)
        if trace:
            zs.append(z.clone())
            epss.append(eps.clone())
        if mode == "ddpm":
            z = ddpm_step(z, eps, t, coef, None if noise is None or t == 0 else noise[k].to(dt))
        else:
            z = ddim_step(z, eps, k, coef)
    if trace:
        return z, zs, epss
    return z


# --------------------------------------------------------------------------------------------
# counter-based Gaussian noise (build-defined; the reference draws from the backend's RNG, :754)
# --------------------------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)
NOISE_TAG = 0x4C444D31
INIT_STEP = 0xFFFFFFFF


# --------------------------------------------------------------------------------------------
# forward process and loss value (training side, no backward)         src/main.rs:584-679
# --------------------------------------------------------------------------------------------
def add_noise_coefficients(num_time: int) -> np.ndarray:
    """[num_time, 2] f32 = {sqrt_alpha, sqrt_one_minus_alpha} as add_noise derives them (:603-624)."""
    ab = get_alphas(num_time)
    out = np.empty((num_time, 2), dtype=np.float32)
    for i in range(num_time):
        a = np.float32(ab[i])
        if np.isnan(a) or np.isinf(a) or a <= 0.0:
            a = np.float32(1e-6)                                                  # :609-610
        else:
            a = np.float32(min(max(a, np.float32(1e-8)), np.float32(0.9999)))     # :612
        out[i, 0] = min(max(np.sqrt(a, dtype=np.float32), np.float32(1e-8)), np.float32(1.0))   # :622
        om = np.float32(1.0) - a                                                  # :623
        out[i, 1] = np.sqrt(np.float32(min(max(om, np.float32(1e-8)), np.float32(1.0))), dtype=np.float32)   # :624
    return out


def add_noise(x: torch.Tensor, time, num_time: int, noise: torch.Tensor) -> torch.Tensor:
    """noisy = x * sqrt_alpha + noise * sqrt_one_minus_alpha (:627), the scalars gathered per sample with
    `(t as usize).min(len - 1)` (:606: a negative i32 becomes a huge usize, i.e. the last entry)."""
    coef = add_noise_coefficients(num_time)
    idx = [int(t) if 0 <= int(t) <= num_time - 1 else num_time - 1 for t in time]
    shape = (len(idx),) + (1,) * (x.dim() - 1)
    sa = torch.from_numpy(coef[idx, 0]).reshape(shape)
    s1m = torch.from_numpy(coef[idx, 1]).reshape(shape)
    return x * sa + noise * s1m


def diffusion_loss(p, z: torch.Tensor, time, num_time: int, noise: torch.Tensor, time_embed_dim: int = 256,
                   train: bool = False, dropout_p: float = 0.5, dropout_seed: int = 0, return_pred: bool = False):
    """DiffusionModel::forward from the latent on (:641-679): add_noise, sin_time_addition, noise_predict,
    mean of the squared difference.  The NaN probes (:645-650, 657-662, 673-677) are not reproduced (D5).
    train=True is how the reference runs it (autodiff backend, :1318): batch-statistics BatchNorm, mid_drop active."""
    z_noisy = add_noise(z, time, num_time, noise)
    t_emb = sin_time_addition(torch.as_tensor(list(time)), time_embed_dim)
    pred = unet_forward(p, z_noisy, t_emb, train=train, dropout_p=dropout_p, dropout_seed=dropout_seed)
    loss = ((pred.flatten(1) - noise.flatten(1)) ** 2).mean().reshape(1)
    return (loss, pred) if return_pred else loss


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """Philox-4x32-10 (Salmon et al. 2011).  ctr [..., 4] uint32, key [2] uint32 -> [..., 4] uint32."""
    c = ctr.astype(np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = c[..., 0].astype(np.uint64) * _PHILOX_M0
        p1 = c[..., 2].astype(np.uint64) * _PHILOX_M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
        n0 = hi1 ^ c[..., 1] ^ k0
        n2 = hi0 ^ c[..., 3] ^ k1
        c = np.stack([n0, lo1, n2, lo0], axis=-1)
        with np.errstate(over="ignore"):
            k0 = np.uint32((int(k0) + int(_PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PHILOX_W1)) & 0xFFFFFFFF)
    return c


def philox_normal_nchw(seed: int, chain0: int, step: int, B: int, C: int, H: int, W: int) -> torch.Tensor:
    """Standard-normal field [B, C, H, W] (f32) used by the in-kernel RNG of the sampler.

    Element (chain b, pixel y,x, channel c): group g = ((y*W + x)*C + c) // 4, lane j = c % 4.
    Philox counter = (g, step, chain0 + b, NOISE_TAG), key = (seed low, seed high).
    Outputs r0..r3 give two Box-Muller pairs: u = ((r >> 8) + 0.5) * 2^-24,
    rad = sqrt(-2 ln u_a), n_a = rad * cos(2 pi u_b), n_b = rad * sin(2 pi u_b);
    lanes 0,1 come from (r0, r1), lanes 2,3 from (r2, r3).
    """
    assert C % 4 == 0
    G = H * W * C // 4
    g = np.arange(G, dtype=np.uint32)
    out = np.empty((B, H, W, C), dtype=np.float32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    for b in range(B):
        ctr = np.stack([g, np.full(G, step & 0xFFFFFFFF, np.uint32),
                        np.full(G, (chain0 + b) & 0xFFFFFFFF, np.uint32),
                        np.full(G, NOISE_TAG, np.uint32)], axis=-1)
        r = philox4x32_10(ctr, key)
        u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
        two_pi = np.float32(6.283185307179586)
        rad_a = np.sqrt(np.float32(-2.0) * np.log(u[:, 0])).astype(np.float32)
        rad_b = np.sqrt(np.float32(-2.0) * np.log(u[:, 2])).astype(np.float32)
        n = np.stack([rad_a * np.cos(two_pi * u[:, 1]), rad_a * np.sin(two_pi * u[:, 1]),
                      rad_b * np.cos(two_pi * u[:, 3]), rad_b * np.sin(two_pi * u[:, 3])], axis=-1)
        out[b] = n.astype(np.float32).reshape(H, W, C)
    return torch.from_numpy(out).permute(0, 3, 1, 2).contiguous()
