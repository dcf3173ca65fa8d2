"""Batch sharding of independent sampling chains over the GPUs of one node.

The reference is single-device.  Chains are independent (inference BatchNorm, no batch statistics),
so the global batch is cut into contiguous slices, one per rank (one process per GPU), every rank
denoises its slice with no communication inside the loop, and ONE all-gather (RCCL over xGMI with the
`nccl` backend) collects the final latents.  Chain b of the global batch always uses generator
stream `chain0 + b`, so the gathered result does not depend on the world size.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def chain_range(global_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[first, last) chains of `rank`: contiguous, sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    base, rem = divmod(global_batch, world_size)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def gather_latents(local: torch.Tensor, global_batch: int, group=None) -> torch.Tensor:
    """All-gathers per-rank latents [b_r, C, H, W] into [global_batch, C, H, W] on every rank.

    With the `gloo` backend tensors are moved through host memory (gloo is given CPU tensors only).
    """
    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != global_batch:
            raise ValueError("no process group and the local slice is not the whole batch")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    sizes = [chain_range(global_batch, r, world)[1] - chain_range(global_batch, r, world)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} chains, expected {sizes[rank]}")
    dev = local.device
    src = local.contiguous()
    if backend == "gloo" and src.is_cuda:
        src = src.cpu()      # synchronous copy
    if len(set(sizes)) == 1:
        out = torch.empty((global_batch,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=group)
    else:
        mx = max(sizes)
        pad = torch.zeros((mx,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[: src.shape[0]] = src
        buf = torch.empty((world * mx,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(buf, pad, group=group)
        out = torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)
    return out.to(dev)


class PendingGather:
    """A latent all-gather in flight.  `wait()` makes the current stream (GPU tensors) or the caller (CPU
    tensors) wait for it and returns the gathered [global_batch, C, H, W] tensor."""

    def __init__(self, out: torch.Tensor, work, keep):
        self._out, self._work, self._keep = out, work, keep

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        return self._out


def gather_latents_async(local: torch.Tensor, global_batch: int, group=None) -> PendingGather:
    """gather_latents that returns at once: the collective runs on the backend's own stream while the caller
    launches the next chains (RCCL: ordered after the work already queued on the current stream).  Ranks
    must hold equally many chains and, with `gloo`, CPU tensors; any other case is done synchronously."""
    if not dist.is_available() or not dist.is_initialized():
        return PendingGather(gather_latents(local, global_batch, group), None, None)
    world = dist.get_world_size(group)
    sizes = {chain_range(global_batch, r, world)[1] - chain_range(global_batch, r, world)[0] for r in range(world)}
    if len(sizes) != 1 or (dist.get_backend(group) == "gloo" and local.is_cuda):
        return PendingGather(gather_latents(local, global_batch, group), None, None)
    if local.shape[0] * world != global_batch:
        raise ValueError(f"this rank holds {local.shape[0]} chains, expected {global_batch // world}")
    src = local.contiguous()
    out = torch.empty((global_batch,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    work = dist.all_gather_into_tensor(out, src, group=group, async_op=True)
    return PendingGather(out, work, src)


def sample_sharded(sample_fn: Callable[[int, int], torch.Tensor], global_batch: int, group=None,
                   rank: Optional[int] = None, world_size: Optional[int] = None) -> torch.Tensor:
    """Runs `sample_fn(local_batch, chain0)` on this rank's slice and gathers all slices.

    sample_fn is e.g. `lambda b, c0: model.ddim_sample(b, 50, spatial=64, seed=s, chain0=c0)`.
    """
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    first, last = chain_range(global_batch, rank, world_size)
    local = sample_fn(last - first, first)
    return gather_latents(local, global_batch, group)
