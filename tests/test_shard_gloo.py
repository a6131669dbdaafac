"""Multi-process path on CPU (gloo, world_size 2): sharding a batch of independent frame pairs over ranks and
gathering in order equals the unsharded run.  The per-rank compute is the CPU oracle on a tiny model (the
product forward needs a GPU; the sharding logic does not care what `forward` is)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ptlflow_amd.shard import partition, run_sharded


def test_partition_covers_everything():
    for n in (0, 1, 5, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [partition(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _forward(x):  # deterministic per-sample function standing in for the model
    return torch.stack([x.mean(dim=(1, 2)), x.amax(dim=(1, 2))], 1)[:, :, :2, :3].contiguous()


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        images = torch.rand(B, 2, 3, 8, 12)
        out = run_sharded(_forward, images)
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 3, 1])
def test_two_rank_gloo_matches_single_process(B):
    torch.manual_seed(0)
    images = torch.rand(B, 2, 3, 8, 12)
    ref = _forward(images)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(out, ref)
