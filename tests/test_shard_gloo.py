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


def _timing_worker(rank, world, port, q):
    """The N > 1 branch of bench.py (`timed_steps`: barrier, max-over-ranks all-reduce, whole-job throughput) on gloo."""
    import time
    from ptlflow_amd.shard import job_throughput, timed_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = {"n": 0}

        def step():                       # rank 1 is the slow rank: the reported time must be ITS time
            calls["n"] += 1
            time.sleep(0.02 if rank == 0 else 0.06)

        elapsed = timed_steps(step, steps=5, warmup=2)
        q.put((rank, calls["n"], elapsed, job_throughput(8, 5, elapsed)))
    finally:
        dist.destroy_process_group()


def test_timed_steps_takes_max_over_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, e0, v0), (_, n1, e1, v1) = got
    assert n0 == n1 == 7                                  # 2 warm-up + exactly 5 timed steps on every rank
    assert e0 == e1 and e0 >= 5 * 0.06 and e0 < 5 * 0.06 + 0.5    # every rank reports the slowest rank's time
    assert v0 == v1 == pytest.approx(8 * 5 * 2 / e0)      # whole-job units: both ranks' work over the max time


def test_timed_steps_single_process():
    from ptlflow_amd.shard import job_throughput, timed_steps
    n = {"c": 0}
    e = timed_steps(lambda: n.__setitem__("c", n["c"] + 1), steps=4, warmup=1)
    assert n["c"] == 5 and e >= 0 and job_throughput(3, 4, 1.0) == 12
