"""Helpers shared by the tests: error metrics and the tolerances of the parity bar.

Floating-point parity bar (per-step epsilon prediction against the CPU restatement):
relative L2 error <= 1e-4 in f32 and <= 2e-2 in bf16.
"""
import torch

TOL = {"fp32": 1e-4, "bf16": 2e-2}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)


def max_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    mse = ((a - b) ** 2).mean().item()
    peak = (b.max() - b.min()).item()
    if mse == 0:
        return float("inf")
    return 10.0 * torch.log10(torch.tensor(peak * peak / mse)).item()


def sub(weights, prefix):
    """Parameters of one sub-module, names without the prefix."""
    p = prefix + "."
    return {k[len(p):]: v for k, v in weights.items() if k.startswith(p)}
