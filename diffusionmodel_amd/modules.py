"""Python surface of the sampling hot path, named after the reference.

Class and method names follow LDM.py (`Attention`, `ResiduelBlock`, `UNet`; LDM.py:152-219) and the
Rust statement that completes it (`TimeAddition`, `sin_time_addition`, `DiffusionModel.get_betas /
get_alphas / noise_predict / denoising_process`; src/main.rs:278-335, 538-583, 631-633, 681-795).
Every forward runs hand-written HIP kernels through libldm_hip.so; PyTorch only provides device
memory and the stream.  Tensors at this boundary are the reference's: NCHW float32 on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, schedule
from .weights import (LDMConfig, attention_param_shapes, decoder_param_shapes, make_synthetic_weights,
                      resblock_param_shapes, synthetic_from_shapes, time_addition_param_shapes, unet_param_shapes)

_DTYPES = {"bf16": _lib.LDM_BF16, "bfloat16": _lib.LDM_BF16, "fp32": _lib.LDM_F32, "float32": _lib.LDM_F32,
           "f32": _lib.LDM_F32}


def _dtype_code(dtype) -> int:
    if isinstance(dtype, torch.dtype):
        dtype = {torch.bfloat16: "bf16", torch.float32: "fp32"}.get(dtype, str(dtype))
    try:
        return _DTYPES[str(dtype)]
    except KeyError:
        raise ValueError(f"dtype must be 'bf16' or 'fp32', got {dtype!r}") from None


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t: torch.Tensor, name: str, shape: Optional[Tuple[int, ...]] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA (HIP) tensor")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.contiguous()


class _Module:
    """A handle of libldm_hip.so with its parameters uploaded."""

    def __init__(self, kind: int, dtype, in_channels: int, time_embed_dim: int, shapes,
                 weights: Optional[Dict[str, torch.Tensor]], seed: int,
                 channels: Optional[Sequence[int]] = None, scaled_attention: bool = False):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.dtype = _dtype_code(dtype)
        if kind == _lib.LDM_MOD_VAE_DECODER:     # in_channels = latent_dimen, time_embed_dim = image channels
            _lib.check(self._lib.ldm_vae_decoder_create(C.byref(self._h), self.dtype, int(time_embed_dim),
                                                        int(in_channels)))
        else:
            ch = (C.c_int * 3)(*channels) if channels is not None else None
            _lib.check(self._lib.ldm_module_create(C.byref(self._h), kind, self.dtype, int(in_channels),
                                                   int(time_embed_dim), ch, int(bool(scaled_attention))))
        if weights is None:
            weights = synthetic_from_shapes(shapes, seed)
        self.param_shapes = shapes
        for name, shape in shapes.items():
            if name not in weights:
                raise KeyError(f"missing parameter {name!r}")
            w = weights[name].detach().to("cpu", torch.float32).contiguous()
            if tuple(w.shape) != tuple(shape):
                raise ValueError(f"parameter {name!r} has shape {tuple(w.shape)}, expected {tuple(shape)}")
            _lib.check(self._lib.ldm_module_set_param(self._h, name.encode(), C.c_void_p(w.data_ptr()), w.numel()))
        _lib.check(self._lib.ldm_module_finalize(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.ldm_module_destroy(h)
            except Exception:
                pass

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


def sin_time_addition(time: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """Sinusoidal embedding [B, dim] of integer timesteps (src/main.rs:309-335)."""
    if not time.is_cuda:
        raise ValueError("time must be a CUDA (HIP) tensor")
    t = time.to(torch.int32).contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.float32, device=t.device)
    lib = _lib.load()
    _lib.check(lib.ldm_sin_time_addition(_ptr(t), t.numel(), dim, _ptr(out), _stream()))
    return out


class TimeAddition(_Module):
    """Linear(dim, 8 dim) -> ReLU -> Linear(8 dim, dim) (src/main.rs:278-307)."""

    def __init__(self, dim: int, weights=None, seed: int = 0):
        self.dim = dim
        super().__init__(_lib.LDM_MOD_TIME_ADDITION, "fp32", dim, 0, time_addition_param_shapes(dim), weights, seed)

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        t = _dev_f32(t, "t")
        if t.dim() != 2 or t.shape[1] != self.dim:
            raise ValueError(f"t must be [B, {self.dim}]")
        out = torch.empty_like(t)
        _lib.check(self._lib.ldm_time_addition_forward(self._h, _ptr(t), _ptr(out), t.shape[0], _stream()))
        return out


class Attention(_Module):
    """Single-head self-attention over the H*W positions with a residual add (LDM.py:152-182,
    src/main.rs:237-277; the Rust statement is followed where the two differ)."""

    def __init__(self, channels: int, dtype="bf16", weights=None, seed: int = 0):
        self.channels = channels
        super().__init__(_lib.LDM_MOD_ATTENTION, dtype, channels, 0, attention_param_shapes(channels), weights, seed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = _dev_f32(x, "x")
        if x.dim() != 4 or x.shape[1] != self.channels:
            raise ValueError(f"x must be [B, {self.channels}, H, W]")
        out = torch.empty_like(x)
        B, _, H, W = x.shape
        _lib.check(self._lib.ldm_attention_forward(self._h, _ptr(x), _ptr(out), B, H, W, _stream()))
        return out


class ResiduelBlock(_Module):
    """conv3x3-BN-SiLU, + SiLU(Linear(t)), conv3x3-BN, + x, SiLU (LDM.py:183-212, src/main.rs:336-388).
    The class name keeps LDM.py's spelling; `ResidualBlock` is an alias."""

    def __init__(self, channels: int, time_embed: int, dtype="bf16", weights=None, seed: int = 0):
        self.channels, self.time_embed = channels, time_embed
        super().__init__(_lib.LDM_MOD_RESBLOCK, dtype, channels, time_embed,
                         resblock_param_shapes(channels, time_embed), weights, seed)

    def forward(self, x: torch.Tensor, time_embed: torch.Tensor, train: bool = False, return_stats: bool = False):
        """train=True: bn1 / bn2 normalise with the statistics of the batch, as on the reference's autodiff backend
        (src/main.rs:1318); return_stats adds the batch (mean, var), each [2 * channels]: bn1 then bn2."""
        x = _dev_f32(x, "x")
        if x.dim() != 4 or x.shape[1] != self.channels:
            raise ValueError(f"x must be [B, {self.channels}, H, W]")
        B, _, H, W = x.shape
        t = _dev_f32(time_embed, "time_embed", (B, self.time_embed))
        out = torch.empty_like(x)
        if not train:
            _lib.check(self._lib.ldm_resblock_forward(self._h, _ptr(x), _ptr(t), _ptr(out), B, H, W, _stream()))
            return out
        mean = var = None
        if return_stats:
            mean = torch.empty(2 * self.channels, dtype=torch.float32, device=x.device)
            var = torch.empty_like(mean)
        _lib.check(self._lib.ldm_resblock_forward_train(self._h, _ptr(x), _ptr(t), _ptr(out), B, H, W, _ptr(mean), _ptr(var),
                                                        _stream()))
        return (out, mean, var) if return_stats else out


ResidualBlock = ResiduelBlock


class UNet(_Module):
    """Unet::new(in_channels, time_embed_dim) / Unet::forward(x, time) (LDM.py:214-216,
    src/main.rs:417-526).  `time` is the [B, time_embed_dim] sinusoidal embedding."""

    def __init__(self, in_channels: int = 64, time_embed_dim: int = 256, channels: Sequence[int] = (64, 128, 256),
                 scaled_attention: bool = False, dtype="bf16", weights=None, seed: int = 0):
        self.cfg = LDMConfig(in_channels, time_embed_dim, tuple(channels), scaled_attention)
        self.in_channels, self.time_embed_dim = in_channels, time_embed_dim
        if weights is None:
            weights = make_synthetic_weights(self.cfg, seed)
        super().__init__(_lib.LDM_MOD_UNET, dtype, in_channels, time_embed_dim, unet_param_shapes(self.cfg), weights,
                         seed, channels=channels, scaled_attention=scaled_attention)

    def forward(self, x: torch.Tensor, time: torch.Tensor, train: bool = False, dropout_p: float = 0.5,
                dropout_seed: int = 0, return_stats: bool = False):
        """train=False: inference semantics (running statistics folded into the convolutions, no dropout).
        train=True: the forward of the reference's training step (autodiff backend, src/main.rs:1318): every
        ResidualBlock's BatchNorm on the statistics of the batch, mid_drop (p = 0.5, :458) active with a mask from the
        counter-based generator keyed by dropout_seed; return_stats adds the batch (mean, var) of the fourteen
        BatchNorms, each [2 * sum of block channels] in block order, bn1 then bn2."""
        x = _dev_f32(x, "x")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"x must be [B, {self.in_channels}, H, W]")
        B, _, H, W = x.shape
        t = _dev_f32(time, "time", (B, self.time_embed_dim))
        out = torch.empty_like(x)
        if not train:
            _lib.check(self._lib.ldm_unet_forward(self._h, _ptr(x), _ptr(t), _ptr(out), B, H, W, _stream()))
            return out
        mean = var = None
        if return_stats:
            c1, c2, c3 = self.cfg.channels
            mean = torch.empty(2 * (2 * c1 + 2 * c2 + 3 * c3), dtype=torch.float32, device=x.device)
            var = torch.empty_like(mean)
        _lib.check(self._lib.ldm_unet_forward_train(self._h, _ptr(x), _ptr(t), _ptr(out), B, H, W, float(dropout_p),
                                                    dropout_seed & (2 ** 64 - 1), _ptr(mean), _ptr(var), _stream()))
        return (out, mean, var) if return_stats else out


Unet = UNet


class Decoder(_Module):
    """SimpleDecoder::new(output, latent_dimen) / forward (LDM.py:86-150 `Decoder`, src/main.rs:96-170):
    Linear(16 latent_dimen -> 128*16*16) + ReLU, two ConvTranspose 4x4/s2 + BatchNorm + ReLU, a
    ConvTranspose 3x3/s1 and tanh.  z [B, latent_dimen, 4, 4] -> image [B, output, 64, 64] in [-1, 1].
    Inference semantics (running statistics, no dropout)."""

    def __init__(self, output: int = 3, latent_dimen: int = 64, dtype="bf16", weights=None, seed: int = 0):
        self.output, self.latent_dimen = output, latent_dimen
        super().__init__(_lib.LDM_MOD_VAE_DECODER, dtype, latent_dimen, output,
                         decoder_param_shapes(output, latent_dimen), weights, seed)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        z = _dev_f32(z, "z")
        if z.dim() != 4 or tuple(z.shape[1:]) != (self.latent_dimen, 4, 4):
            raise ValueError(f"z must be [B, {self.latent_dimen}, 4, 4]")
        out = torch.empty((z.shape[0], self.output, 64, 64), dtype=torch.float32, device=z.device)
        _lib.check(self._lib.ldm_vae_decode(self._h, _ptr(z), _ptr(out), z.shape[0], _stream()))
        return out

    decode = forward


def to_rgb8(img: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] float32 in [-1,1] -> [B,H,W,C] uint8 with the arithmetic of save_as_image
    (src/main.rs:1297-1308): clamp((v + 1) * 127.5, 0, 255) truncated."""
    img = _dev_f32(img, "img")
    if img.dim() != 4:
        raise ValueError("img must be [B, C, H, W]")
    B, Cc, H, W = img.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=img.device)
    _lib.check(_lib.load().ldm_to_rgb8(_ptr(img), _ptr(out), B, Cc, H, W, _stream()))
    return out


def sched_step(z: torch.Tensor, eps: torch.Tensor, coef5, mode: str = "ddpm", last: bool = False,
               noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One reverse-process update of `z` in place (src/main.rs:712-756,765); returns z."""
    z = _dev_f32(z, "z")
    eps = _dev_f32(eps, "eps", tuple(z.shape))
    if noise is not None:
        noise = _dev_f32(noise, "noise", tuple(z.shape))
    c = (C.c_float * 5)(*[float(v) for v in coef5])
    lib = _lib.load()
    _lib.check(lib.ldm_sched_step(_ptr(z), _ptr(eps), _ptr(noise), z.numel(), c,
                                  _lib.LDM_DDIM if mode == "ddim" else _lib.LDM_DDPM, int(last), _stream()))
    return z


def latent_is_finite(t: torch.Tensor) -> bool:
    """The reference's probe `mean().is_nan() || mean().is_infinite()` (src/main.rs:780-781, 788-789, 647-648) as one
    reduction over the elements on the device: True when every element of `t` is finite.  Reads one flag back (the
    host synchronisation the reference has at the same place)."""
    t = _dev_f32(t, "t")
    flag = torch.empty(1, dtype=torch.int32, device=t.device)
    _lib.check(_lib.load().ldm_latent_is_finite(_ptr(t), t.numel(), _ptr(flag), _stream()))
    return bool(flag.item())


def philox_normal(batch: int, channels: int, H: int, W: int, seed: int, chain0: int = 0, step: int = 0xFFFFFFFF,
                  device="cuda") -> torch.Tensor:
    """The Gaussian field the sampler's built-in generator draws at `step` (0xFFFFFFFF: initial latent)."""
    out = torch.empty((batch, channels, H, W), dtype=torch.float32, device=device)
    lib = _lib.load()
    _lib.check(lib.ldm_philox_normal(_ptr(out), batch, channels, H, W, seed & (2 ** 64 - 1), chain0, step & 0xFFFFFFFF,
                                     _stream()))
    return out


class DiffusionModel:
    """DiffusionModel::new(latent_dimen, in_channels, num_time, time_embed_dim) (src/main.rs:538-551).
    `denoising_process` returns the final latent (what the reference hands to vae.decode at :786);
    with `decoder_weights` given, `decode` / `sample_images` add the reference's last step (Vae::decode,
    :786) for 4x4 latents.  The VAE encoder stays with the reference."""

    def __init__(self, latent_dimen: int = 64, in_channels: int = 3, num_time: int = 50, time_embed_dim: int = 256,
                 channels: Sequence[int] = (64, 128, 256), scaled_attention: bool = False, dtype="bf16",
                 weights=None, seed: int = 0, device="cuda", decoder_weights=None):
        self.latent_dimen, self.in_channels = latent_dimen, in_channels
        self.num_time, self.time_embed_dim = num_time, time_embed_dim
        self.device = torch.device(device)
        self.unet = UNet(latent_dimen, time_embed_dim, channels, scaled_attention, dtype, weights, seed)
        self.decoder = (Decoder(in_channels, latent_dimen, dtype, decoder_weights)
                        if decoder_weights is not None else None)
        self._lib = _lib.load()
        self._samplers: Dict[Tuple[int, int, int], C.c_void_p] = {}

    def __del__(self):
        for h in getattr(self, "_samplers", {}).values():
            try:
                self._lib.ldm_sampler_destroy(h)
            except Exception:
                pass
        self._samplers = {}

    # -- schedule -------------------------------------------------------------------------------
    def get_betas(self) -> torch.Tensor:
        return torch.from_numpy(schedule.get_betas(self.num_time))

    def get_alphas(self) -> torch.Tensor:
        """Returns alpha-bar (the cumulative product), as the reference's get_alphas does."""
        return torch.from_numpy(schedule.get_alphas(self.num_time))

    def ddpm_coefficients(self) -> torch.Tensor:
        return torch.from_numpy(schedule.ddpm_coefficients(self.num_time))

    def ddim_timesteps(self, steps: int = 50):
        return schedule.ddim_timesteps(self.num_time, steps)

    def ddim_coefficients(self, steps: int = 50) -> torch.Tensor:
        return torch.from_numpy(schedule.ddim_coefficients(self.num_time, steps))

    # -- network --------------------------------------------------------------------------------
    def noise_predict(self, x_t: torch.Tensor, time_emb: torch.Tensor, train: bool = False, dropout_p: float = 0.5,
                      dropout_seed: int = 0) -> torch.Tensor:
        return self.unet.forward(x_t, time_emb, train=train, dropout_p=dropout_p, dropout_seed=dropout_seed)

    # -- forward process and loss value (the step before the path in training; no backward here) ----
    def add_noise(self, x: torch.Tensor, time: torch.Tensor, noise: Optional[torch.Tensor] = None):
        """DiffusionModel::add_noise(x, time) (src/main.rs:584-630): returns (noisy, noise) with
        noisy = x * sqrt(alpha_bar_t) + noise * sqrt(1 - alpha_bar_t) per sample (the reference's clamps
        included).  `noise` may be injected; otherwise it is drawn with torch on the device."""
        x = _dev_f32(x, "x")
        if x.dim() < 2 or (x.numel() // x.shape[0]) % 4 != 0:
            raise ValueError("x must be [B, ...] with a sample size that is a multiple of 4")
        B = x.shape[0]
        if noise is None:
            noise = torch.randn_like(x)
        noise = _dev_f32(noise, "noise", tuple(x.shape))
        t = time.to(device=x.device, dtype=torch.int32).contiguous()
        if t.numel() != B:
            raise ValueError(f"time must hold {B} timesteps")
        if getattr(self, "_an_coef", None) is None or self._an_coef.device != x.device:
            self._an_coef = torch.from_numpy(schedule.add_noise_coefficients(self.num_time)).to(x.device)
        out = torch.empty_like(x)
        _lib.check(self._lib.ldm_add_noise(_ptr(x), _ptr(noise), _ptr(t), _ptr(self._an_coef), self.num_time, _ptr(out),
                                           B, x.numel() // B, _stream()))
        return out, noise

    def forward(self, z: torch.Tensor, time: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None, check_finite: bool = False, train: bool = True,
                dropout_p: float = 0.5, dropout_seed: int = 0) -> torch.Tensor:
        """The loss value of DiffusionModel::forward (src/main.rs:635-679) from the latent on: random
        timesteps (or `time`), add_noise, sinusoidal embedding, noise prediction, mean squared error.
        z is what the reference obtains from vae.encode(x).detach() (:636; the encoder stays with the
        reference).  Returns a tensor of shape [1].  Value only: there is no backward pass here.
        train=True (the default: the reference only ever calls this on its autodiff backend, :1318, :1554) predicts
        the noise with batch-statistics BatchNorm and mid_drop active (mask keyed by dropout_seed); train=False
        evaluates the same loss with the sampler's inference semantics.
        check_finite: the reference's three probes (:647-651, 657-662, 672-676): a noisy latent, a prediction or a
        loss that is not finite gives the sentinel loss 100.0 (each probe reads a flag back, as the reference's do)."""
        z = _dev_f32(z, "z")
        B = z.shape[0]
        if time is None:
            time = torch.randint(0, self.num_time, (B,), device=z.device)
        z_noisy, noise = self.add_noise(z, time, noise)
        sentinel = lambda: torch.full((1,), 100.0, dtype=torch.float32, device=z.device)   # noqa: E731
        if check_finite and not latent_is_finite(z_noisy):
            return sentinel()
        t_emb = sin_time_addition(time.to(z.device), self.time_embed_dim)
        pred = self.noise_predict(z_noisy, t_emb, train=train, dropout_p=dropout_p, dropout_seed=dropout_seed)
        if check_finite and not latent_is_finite(pred):
            return sentinel()
        ws = torch.empty(1024, dtype=torch.float32, device=z.device)
        out = torch.empty(4, dtype=torch.float32, device=z.device)[:1]     # 16-byte aligned for the probe
        _lib.check(self._lib.ldm_mse(_ptr(pred), _ptr(noise), pred.numel(), _ptr(ws), _ptr(out), _stream()))
        if check_finite and not latent_is_finite(out):
            return sentinel()
        return out

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """vae.decode(z) (src/main.rs:786): [B, latent_dimen, 4, 4] -> [B, in_channels, 64, 64]."""
        if self.decoder is None:
            raise RuntimeError("this DiffusionModel was built without decoder_weights")
        return self.decoder.forward(z)

    def sample_images(self, batch_size: int, check_finite: bool = False, **kw) -> torch.Tensor:
        """denoising_process followed by vae.decode, as the reference's denoising_process returns
        (src/main.rs:681-795): [B, in_channels, 64, 64] in [-1, 1].  check_finite: the reference's two probes after
        the loop: a latent that is not finite is replaced by zeros before decoding (:780-784), a decoded image that
        is not finite by zeros (:788-792)."""
        kw.pop("return_trace", None)
        img = self.decode(self.denoising_process(batch_size, spatial=4, check_finite=check_finite, **kw))
        if check_finite and not latent_is_finite(img):
            img.zero_()
        return img

    # -- sampling -------------------------------------------------------------------------------
    def _sampler(self, B: int, H: int, W: int) -> C.c_void_p:
        key = (B, H, W)
        if key not in self._samplers:
            h = C.c_void_p()
            _lib.check(self._lib.ldm_sampler_create(C.byref(h), self.unet._h, self.num_time, B, H, W))
            self._samplers[key] = h
        return self._samplers[key]

    def _run(self, mode: int, steps: int, batch_size: int, spatial, z0, noise, seed: int, chain0: int, use_graph: bool,
             return_trace: bool, check_finite: bool = False):
        H, W = (spatial, spatial) if isinstance(spatial, int) else spatial
        Cl = self.latent_dimen
        if z0 is not None:
            z0 = _dev_f32(z0, "z0", (batch_size, Cl, H, W))
        if noise is not None:
            noise = _dev_f32(noise, "noise", (steps, batch_size, Cl, H, W))
        dev = z0.device if z0 is not None else self.device
        out = torch.empty((batch_size, Cl, H, W), dtype=torch.float32, device=dev)
        zt = et = None
        if return_trace:
            zt = torch.empty((steps, batch_size, Cl, H, W), dtype=torch.float32, device=dev)
            et = torch.empty_like(zt)
        h = self._sampler(batch_size, H, W)
        _lib.check(self._lib.ldm_sampler_run(h, mode, steps, _ptr(z0), _ptr(noise), seed & (2 ** 64 - 1),
                                             chain0 & 0xFFFFFFFF, _ptr(out), _ptr(zt), _ptr(et), int(use_graph),
                                             _stream()))
        if check_finite and not latent_is_finite(out):     # src/main.rs:780-784: the fallback tensor is all zeros
            out.zero_()
        if return_trace:
            return out, zt, et
        return out

    def denoising_process(self, batch_size: int, spatial=4, z0: Optional[torch.Tensor] = None,
                          noise: Optional[torch.Tensor] = None, seed: int = 0, chain0: int = 0,
                          use_graph: bool = True, return_trace: bool = False, check_finite: bool = False):
        """DDPM ancestral sampling, i = num_time-1 .. 0 (src/main.rs:681-777).

        spatial: latent height/width (the reference hard-codes 4x4, :683).  z0 / noise: optional
        injected initial latent [B,C,H,W] and per-step noise [num_time,B,C,H,W]; otherwise the built-in
        counter-based generator is used with (seed, chain0).  check_finite: the reference's probe after the loop
        (:780-784): a final latent that is not finite is replaced by zeros (one flag is read back; the per-step
        probes of the reference are not reproduced)."""
        return self._run(_lib.LDM_DDPM, self.num_time, batch_size, spatial, z0, noise, seed, chain0, use_graph,
                         return_trace, check_finite)

    def ddim_sample(self, batch_size: int, steps: int = 50, spatial=4, z0: Optional[torch.Tensor] = None,
                    seed: int = 0, chain0: int = 0, use_graph: bool = True, return_trace: bool = False,
                    check_finite: bool = False):
        """Deterministic DDIM (eta = 0) over `steps` strided timesteps; not part of the reference."""
        return self._run(_lib.LDM_DDIM, steps, batch_size, spatial, z0, None, seed, chain0, use_graph, return_trace,
                         check_finite)
