"""world_size-2 gloo test (CPU) of the N>1 path's host logic: rank sharding and the ragged mask all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from walkgpt_amd.dist import all_gather_masks, all_gather_masks_uniform, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _masks_for(rank):
    g = torch.Generator().manual_seed(100 + rank)
    if rank == 0:  # two images, ragged T and sizes
        return [torch.randn(2, 5, 7, generator=g), torch.randn(1, 3, 4, generator=g)]
    return [torch.randn(3, 6, 2, generator=g)]  # one image


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = all_gather_masks(_masks_for(rank))
        ok = len(got) == world
        for r in range(world):
            want = _masks_for(r)
            ok = ok and len(got[r]) == len(want) and all(torch.equal(a, b) for a, b in zip(got[r], want))
        # a rank without a single mask (a shard with no [SEG] token) still joins the collectives
        mine = [] if rank == 1 else [torch.full((2, 4, 3), 7.0)]
        got2 = all_gather_masks(mine)
        ok = ok and len(got2[1]) == 0 and len(got2[0]) == 1 and bool((got2[0][0] == 7.0).all()) and got2[0][0].shape == (2, 4, 3)
        u = all_gather_masks_uniform(torch.full((2, 3, 3), float(rank)))
        ok = ok and u.shape == (4, 3, 3) and bool((u[:2] == 0).all()) and bool((u[2:] == 1).all())
        # bf16 on the wire (2 bytes per logit): signs, hence thresholded masks, are those of the fp32 logits
        got3 = all_gather_masks(_masks_for(rank), wire_dtype=torch.bfloat16)
        for r in range(world):
            want = _masks_for(r)
            ok = ok and all(a.dtype == torch.bfloat16 and torch.equal(a, b.to(torch.bfloat16)) and torch.equal(a > 0, b > 0)
                            for a, b in zip(got3[r], want))
        out = torch.empty(4, 3, 3, dtype=torch.bfloat16)
        u2 = all_gather_masks_uniform(torch.full((2, 3, 3), rank - 0.5), out=out, wire_dtype=torch.bfloat16)
        ok = ok and u2 is out and bool((out[:2] == -0.5).all()) and bool((out[2:] == 0.5).all())
        try:
            all_gather_masks_uniform(torch.zeros(2, 3, 3), out=torch.empty(4, 3, 3), wire_dtype=torch.bfloat16)   # fp32 buffer, bf16 payload
            ok = False
        except ValueError:
            pass
        # images sharded by rank cover the batch exactly once
        a, b = shard_range(9, rank, world)
        t = torch.zeros(9)
        t[a:b] = 1
        dist.all_reduce(t)
        ok = ok and bool((t == 1).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_ragged_mask_all_gather_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(2))
    assert results == {0: True, 1: True}
