"""ctypes binding of libldm_hip.so (the C ABI declared in include/ldm_hip.h).

There is no fallback: if the library is missing, or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

LDM_OK = 0
LDM_F32, LDM_BF16 = 0, 1
LDM_DDPM, LDM_DDIM = 0, 1
LDM_MOD_UNET, LDM_MOD_RESBLOCK, LDM_MOD_ATTENTION, LDM_MOD_TIME_ADDITION, LDM_MOD_VAE_DECODER = 0, 1, 2, 3, 4

_LIB_NAME = "libldm_hip.so"
_lib: Optional[C.CDLL] = None

_vp, _i, _l, _u64, _u32 = C.c_void_p, C.c_int, C.c_long, C.c_ulonglong, C.c_uint
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every function of include/ldm_hip.h
SIGNATURES = {
    "ldm_last_error": (C.c_char_p, []),
    "ldm_version": (_i, []),
    "ldm_get_betas": (_i, [_i, _fp]),
    "ldm_get_alphas": (_i, [_i, _fp]),
    "ldm_ddpm_coefficients": (_i, [_i, _fp]),
    "ldm_ddim_timesteps": (_i, [_i, _i, _ip]),
    "ldm_ddim_coefficients": (_i, [_i, _i, _fp]),
    "ldm_sin_time_addition": (_i, [_vp, _i, _i, _vp, _vp]),
    "ldm_module_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _ip, _i]),
    "ldm_module_set_param": (_i, [_vp, C.c_char_p, _vp, _l]),
    "ldm_module_finalize": (_i, [_vp]),
    "ldm_module_destroy": (None, [_vp]),
    "ldm_unet_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ldm_resblock_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ldm_unet_forward_train": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, _u64, _vp, _vp, _vp]),
    "ldm_resblock_forward_train": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ldm_batchnorm_ws_floats": (_l, [_i, _l, _i]),
    "ldm_batchnorm_act": (_i, [_i, _vp, _vp, _vp, _i, _l, _i, _vp, _vp, C.c_float, _i, _vp, C.c_float, _u64, _vp, _vp, _vp, _vp]),
    "ldm_attention_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ldm_time_addition_forward": (_i, [_vp, _vp, _vp, _i, _vp]),
    "ldm_add_noise_coefficients": (_i, [_i, _fp]),
    "ldm_add_noise": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _l, _vp]),
    "ldm_mse": (_i, [_vp, _vp, _l, _vp, _vp, _vp]),
    "ldm_mse_noalloc": (_i, [_vp, _vp, _l, _vp, _vp]),
    "ldm_mse_noalloc_prepare": (_i, []),
    "ldm_latent_is_finite": (_i, [_vp, _l, _vp, _vp]),
    "ldm_last_kernel": (C.c_char_p, []),
    "ldm_vae_decoder_create": (_i, [C.POINTER(_vp), _i, _i, _i]),
    "ldm_vae_decode": (_i, [_vp, _vp, _vp, _i, _vp]),
    "ldm_to_rgb8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ldm_conv_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp]),
    "ldm_conv_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ldm_conv_destroy": (None, [_vp]),
    "ldm_sched_step": (_i, [_vp, _vp, _vp, _l, _fp, _i, _i, _vp]),
    "ldm_sampler_create": (_i, [C.POINTER(_vp), _vp, _i, _i, _i, _i]),
    "ldm_sampler_run": (_i, [_vp, _i, _i, _vp, _vp, _u64, _u32, _vp, _vp, _vp, _i, _vp]),
    "ldm_sampler_destroy": (None, [_vp]),
    "ldm_philox_normal": (_i, [_vp, _i, _i, _i, _i, _u64, _u32, _u32, _vp]),
}


def lib_path() -> str:
    # LDM_HIP_LIB: developer override to load another build of the same library
    return os.environ.get("LDM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load() -> C.CDLL:
    """Loads the library once and declares every signature."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C diffusionmodel_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class LdmError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != LDM_OK:
        msg = load().ldm_last_error()
        raise LdmError(f"libldm_hip error {rc}: {msg.decode() if msg else '?'}")
