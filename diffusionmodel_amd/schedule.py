"""Noise schedule and reverse-step scalars (host side, f32), computed by libldm_hip.so in the
reference's operation order (src/main.rs:553-583, 712-747).  No GPU is needed for these."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

from . import _lib

_FP = C.POINTER(C.c_float)


def get_betas(num_time: int) -> np.ndarray:
    out = np.empty(num_time, dtype=np.float32)
    _lib.check(_lib.load().ldm_get_betas(num_time, out.ctypes.data_as(_FP)))
    return out


def get_alphas(num_time: int) -> np.ndarray:
    """alpha-bar (running product of 1 - beta): what the reference's get_alphas returns."""
    out = np.empty(num_time, dtype=np.float32)
    _lib.check(_lib.load().ldm_get_alphas(num_time, out.ctypes.data_as(_FP)))
    return out


def ddpm_coefficients(num_time: int) -> np.ndarray:
    out = np.empty((num_time, 5), dtype=np.float32)
    _lib.check(_lib.load().ldm_ddpm_coefficients(num_time, out.ctypes.data_as(_FP)))
    return out


def ddim_timesteps(num_time: int, steps: int) -> List[int]:
    out = np.empty(steps, dtype=np.int32)
    _lib.check(_lib.load().ldm_ddim_timesteps(num_time, steps, out.ctypes.data_as(C.POINTER(C.c_int))))
    return [int(v) for v in out]


def ddim_coefficients(num_time: int, steps: int) -> np.ndarray:
    out = np.empty((steps, 5), dtype=np.float32)
    _lib.check(_lib.load().ldm_ddim_coefficients(num_time, steps, out.ctypes.data_as(_FP)))
    return out


def add_noise_coefficients(num_time: int) -> np.ndarray:
    """[num_time, 2] = {sqrt_alpha, sqrt_one_minus_alpha} with the clamps of add_noise (src/main.rs:603-624)."""
    out = np.empty((num_time, 2), dtype=np.float32)
    _lib.check(_lib.load().ldm_add_noise_coefficients(num_time, out.ctypes.data_as(_FP)))
    return out
