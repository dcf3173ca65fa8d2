"""Parameter inventory of the UNet side of the reference and seeded synthetic weights.

Names mirror the Rust struct fields (src/main.rs:238-243, 279-282, 337-343, 390-415) so a
checkpoint exported from the reference maps one to one.  Layouts are burn's:
Conv2d [Cout,Cin,kh,kw], ConvTranspose2d [Cin,Cout,kh,kw], Linear [d_in,d_out],
BatchNorm gamma/beta/running_mean/running_var [C].

No checkpoint ships with the reference, so every run uses seeded synthetic weights.  The
initialisation here is variance preserving (the reference's KaimingUniform{gain: 0.1},
src/main.rs:41-44, makes activations collapse towards zero, which would make relative-error
checks vacuous); BatchNorm running statistics are randomised so that the fold is exercised.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import torch


@dataclass(frozen=True)
class LDMConfig:
    """Shape of the UNet: `in_channels` is the latent channel count (Unet::new's in_channels,
    src/main.rs:418), `channels` the three pyramid widths (64/128/256 in the reference)."""
    in_channels: int = 64
    time_embed_dim: int = 256
    channels: Tuple[int, int, int] = (64, 128, 256)
    scaled_attention: bool = False      # extra Attention(channels[1]) after up_res1

    @staticmethod
    def scaled() -> "LDMConfig":
        return LDMConfig(64, 256, (128, 256, 512), True)


def _resblock(prefix: str, c: int, te: int, out: "OrderedDict[str, Tuple[int, ...]]") -> None:
    for k in ("1", "2"):
        out[f"{prefix}.conv{k}.weight"] = (c, c, 3, 3)
        out[f"{prefix}.conv{k}.bias"] = (c,)
        for s in ("gamma", "beta", "running_mean", "running_var"):
            out[f"{prefix}.bn{k}.{s}"] = (c,)
    out[f"{prefix}.time.weight"] = (te, c)
    out[f"{prefix}.time.bias"] = (c,)


def _attention(prefix: str, c: int, out) -> None:
    for n in ("question", "key", "val", "out"):
        out[f"{prefix}.{n}.weight"] = (c, c)
        out[f"{prefix}.{n}.bias"] = (c,)


def resblock_param_shapes(channels: int, time_embed: int, prefix: str = "") -> "OrderedDict[str, Tuple[int, ...]]":
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _resblock(prefix.rstrip("."), channels, time_embed, out)
    return OrderedDict((k.lstrip("."), v) for k, v in out.items())


def attention_param_shapes(channels: int) -> "OrderedDict[str, Tuple[int, ...]]":
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _attention("", channels, out)
    return OrderedDict((k.lstrip("."), v) for k, v in out.items())


def time_addition_param_shapes(dim: int) -> "OrderedDict[str, Tuple[int, ...]]":
    return OrderedDict([("linear1.weight", (dim, dim * 8)), ("linear1.bias", (dim * 8,)),
                        ("linear2.weight", (dim * 8, dim)), ("linear2.bias", (dim,))])


def unet_param_shapes(cfg: LDMConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    c1, c2, c3 = cfg.channels
    te, cl = cfg.time_embed_dim, cfg.in_channels
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    out["time_emb.linear1.weight"] = (te, te * 8)
    out["time_emb.linear1.bias"] = (te * 8,)
    out["time_emb.linear2.weight"] = (te * 8, te)
    out["time_emb.linear2.bias"] = (te,)
    out["down1.weight"] = (c1, cl, 3, 3); out["down1.bias"] = (c1,)
    _resblock("down_res1", c1, te, out)
    out["down2.weight"] = (c2, c1, 3, 3); out["down2.bias"] = (c2,)
    _resblock("down_res2", c2, te, out)
    out["down3.weight"] = (c3, c2, 3, 3); out["down3.bias"] = (c3,)
    _resblock("down_res3", c3, te, out)
    _resblock("mid_res1", c3, te, out)
    _attention("mid_atten", c3, out)
    _resblock("mid_res2", c3, te, out)
    out["up1.weight"] = (c3, c2, 4, 4); out["up1.bias"] = (c2,)
    out["up_conv1.weight"] = (c2, 2 * c2, 1, 1); out["up_conv1.bias"] = (c2,)
    _resblock("up_res1", c2, te, out)
    if cfg.scaled_attention:
        _attention("up_atten", c2, out)
    out["up2.weight"] = (c2, c1, 4, 4); out["up2.bias"] = (c1,)
    out["up_conv2.weight"] = (c1, 2 * c1, 1, 1); out["up_conv2.bias"] = (c1,)
    _resblock("up_res2", c1, te, out)
    out["out_conv.weight"] = (cl, c1, 3, 3); out["out_conv.bias"] = (cl,)
    return out


def decoder_param_shapes(output: int = 3, latent_dimen: int = 64) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameters of SimpleDecoder (src/main.rs:96-147) by Rust field path; ConvTranspose2d weights are
    [Cin,Cout,kh,kw], the Linear's [d_in,d_out]."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    out["transfer.weight"] = (latent_dimen * 16, 128 * 16 * 16)
    out["transfer.bias"] = (128 * 16 * 16,)
    out["reverse1.weight"] = (128, 64, 4, 4); out["reverse1.bias"] = (64,)
    for s in ("gamma", "beta", "running_mean", "running_var"):
        out[f"bn_1.{s}"] = (64,)
    out["reverse2.weight"] = (64, 32, 4, 4); out["reverse2.bias"] = (32,)
    for s in ("gamma", "beta", "running_mean", "running_var"):
        out[f"bn_2.{s}"] = (32,)
    out["reverse3.weight"] = (32, output, 3, 3); out["reverse3.bias"] = (output,)
    return out


def encoder_param_shapes(in_channels: int = 3, latent_dimen: int = 64) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameters of SimpleEncoder (src/main.rs:22-74) by Rust field path.  The encoder is not run by this
    package; its shapes are needed to write a complete DiffusionModel record."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for name, bn, cin, cout in (("conv1", "bn_1", in_channels, 32), ("conv2", "bn_2", 32, 64), ("conv3", "bn_3", 64, 128)):
        out[f"{name}.weight"] = (cout, cin, 3, 3); out[f"{name}.bias"] = (cout,)
        for s in ("gamma", "beta", "running_mean", "running_var"):
            out[f"{bn}.{s}"] = (cout,)
    for name in ("mean", "variance"):
        out[f"{name}.weight"] = (128 * 16 * 16, latent_dimen * 16); out[f"{name}.bias"] = (latent_dimen * 16,)
    return out


def synthetic_from_shapes(shapes: "Dict[str, Tuple[int, ...]]", seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded f32 CPU tensors for the given name -> shape map (see the module docstring)."""
    g = torch.Generator().manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def uni(shape, a):
        return (torch.rand(shape, generator=g) * 2.0 - 1.0) * a

    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight":
            if len(shape) == 4:
                # fan-in of a Conv2d [Co,Ci,kh,kw]; a ConvTranspose2d [Ci,Co,4,4] at stride 2 sums
                # Ci * 4 products per output pixel
                is_convT = shape[2] == 4
                fan_in = shape[0] * 4 if is_convT else shape[1] * shape[2] * shape[3]
                if name.rsplit(".", 2)[-2] == "reverse3":      # ConvTranspose2d [Ci,Co,3,3] at stride 1
                    fan_in = shape[0] * 9
            else:
                fan_in = shape[0]
            out[name] = uni(shape, math.sqrt(3.0 / fan_in))
        elif leaf == "bias":
            out[name] = torch.randn(shape, generator=g) * 0.05
        elif leaf == "gamma":
            out[name] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif leaf == "beta":
            out[name] = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_mean":
            out[name] = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            out[name] = 0.5 + torch.rand(shape, generator=g)
        else:
            raise ValueError(name)
        out[name] = out[name].to(torch.float32).contiguous()
    return out


def make_synthetic_weights(cfg: LDMConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return synthetic_from_shapes(unet_param_shapes(cfg), seed)
