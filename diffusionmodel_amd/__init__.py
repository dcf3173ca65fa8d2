"""MI355X-native sampling hot path of the ragu812/diffusionModel latent-diffusion model.

Same class and method names as the reference (LDM.py / src/main.rs); hand-written HIP kernels for
gfx950 underneath (libldm_hip.so, C ABI in include/ldm_hip.h).
"""
from .weights import LDMConfig, decoder_param_shapes, make_synthetic_weights, synthetic_from_shapes, unet_param_shapes
from .modules import (Attention, Decoder, DiffusionModel, ResidualBlock, ResiduelBlock, TimeAddition, UNet, Unet,
                      latent_is_finite, philox_normal, sched_step, sin_time_addition, to_rgb8)
from .mpk import MpkFormatError, load_model, load_model_weights, read_mpk, save_model_weights
from .parallel import chain_range, gather_latents, gather_latents_async, sample_sharded

__all__ = ["MpkFormatError", "load_model", "load_model_weights", "read_mpk", "save_model_weights", "Attention", "Decoder", "to_rgb8", "decoder_param_shapes", "synthetic_from_shapes", "DiffusionModel", "ResidualBlock", "ResiduelBlock", "TimeAddition", "UNet", "Unet",
           "sin_time_addition", "sched_step", "philox_normal", "latent_is_finite", "LDMConfig", "make_synthetic_weights",
           "unet_param_shapes", "chain_range", "gather_latents", "gather_latents_async", "sample_sharded"]
