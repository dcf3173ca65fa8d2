"""The N > 1 path on CPU: two processes over gloo shard the chains and all-gather the latents."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sample(local_batch, chain0, C=4, S=2):
    """Stand-in for model.ddim_sample: a deterministic function of the GLOBAL chain index only."""
    out = torch.empty(local_batch, C, S, S)
    for b in range(local_batch):
        g = torch.Generator().manual_seed(1000 + chain0 + b)
        out[b] = torch.randn(C, S, S, generator=g)
    return out


def _worker(rank, world, port, global_batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffusionmodel_amd import chain_range, sample_sharded
        got = sample_sharded(_fake_sample, global_batch)
        first, last = chain_range(global_batch, rank, world)
        q.put((rank, got, (first, last)))
        # the same gather left in flight while the next batch is produced (what bench.py does between calls)
        from diffusionmodel_amd import gather_latents_async
        pend = [gather_latents_async(_fake_sample(last - first, first + 100 * k), global_batch) for k in range(3)]
        for k, pg in enumerate(pend):
            full = pg.wait()
            if global_batch % world == 0:
                assert torch.equal(full, torch.cat([_fake_sample(global_batch // world, r * (global_batch // world) + 100 * k)
                                                    for r in range(world)]))
            else:
                assert full.shape[0] == global_batch
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 7])
def test_two_rank_gather_equals_single_process(global_batch):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _fake_sample(global_batch, 0)
    ranges = sorted(r[2] for r in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == global_batch and ranges[0][1] == ranges[1][0]
    for _, got, _ in results:
        assert got.shape == ref.shape
        assert torch.equal(got, ref)          # bit-exact: independent chains, per-chain streams


def test_single_process_passthrough():
    from diffusionmodel_amd import gather_latents, sample_sharded
    x = torch.randn(3, 2, 2, 2)
    assert gather_latents(x, 3) is x
    with pytest.raises(ValueError):
        gather_latents(x, 4)
    assert torch.equal(sample_sharded(_fake_sample, 5), _fake_sample(5, 0))
