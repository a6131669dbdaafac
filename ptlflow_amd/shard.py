"""Batch sharding of independent frame pairs over the GPUs of one node.

One process per GPU (launched by `torch.distributed.run`), every rank holds a full model replica (21 MB of
weights) and takes a contiguous slice of the batch; there is **no data-path collective** — frame pairs are
independent units (`B` is the leading dim of every tensor on the path, raft/corr.py:22).  The only
communication is gathering the results (and, in bench.py, a barrier + max-time reduce).  With warm start
(base_model.py:395-428) consecutive frames of one video are sequentially dependent, so shard by *sequence*:
`partition` works on any unit count.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def partition(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `n_units` for `rank`; the first `n_units % world` ranks get one extra."""
    if not (0 <= rank < world) or n_units < 0:
        raise ValueError("bad partition arguments")
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(images: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """`images` [B, 2, 3, H, W] -> this rank's slice (may be empty when B < world)."""
    a, b = partition(images.shape[0], rank, world)
    return images[a:b]


def run_sharded(forward: Callable[[torch.Tensor], torch.Tensor], images: torch.Tensor,
                gather: bool = True) -> torch.Tensor:
    """Run `forward` on this rank's share of `images` and (optionally) all-gather the flows in batch order.

    Works with any initialised process group (RCCL on GPUs, gloo on CPU); ragged shares are handled by
    gathering python objects' sizes first.  Without a process group it is just `forward(images)`."""
    if not (dist.is_available() and dist.is_initialized()):
        return forward(images)
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_batch(images, rank, world)
    out = forward(mine) if mine.shape[0] > 0 else None
    if not gather:
        return out
    shape_tail = [None]
    if out is not None:
        shape_tail = [tuple(out.shape[1:])]
    tails: List = [None] * world
    dist.all_gather_object(tails, shape_tail[0])
    tail = next(t for t in tails if t is not None)
    parts = []
    for r in range(world):
        a, b = partition(images.shape[0], r, world)
        buf = torch.empty((b - a, *tail), dtype=torch.float32, device=images.device if out is None else out.device)
        if r == rank and out is not None:
            buf.copy_(out)
        if b > a:
            dist.broadcast(buf, src=r)
        parts.append(buf)
    return torch.cat(parts, 0)


def timed_steps(step: Callable[[], object], steps: int, warmup: int, sync: Callable[[], None] = lambda: None):
    """bench.py's timing protocol, one place for 1 and N ranks: `warmup` untimed steps, then EXACTLY `steps` timed steps
    bracketed by (device sync, barrier, device sync) on both sides; the elapsed time is the MAX over ranks (all-reduce),
    so `units / elapsed` is the whole job's throughput.  `sync` is the device synchronisation (torch.cuda.synchronize on
    GPUs, a no-op on CPU); any initialised process group works (RCCL on GPUs, gloo on CPU).  Returns seconds."""
    import time
    grouped = dist.is_available() and dist.is_initialized()
    for _ in range(warmup):
        step()
    sync()
    if grouped:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if grouped:
        backend = dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def job_throughput(units_per_rank_per_step: int, steps: int, elapsed: float) -> float:
    """Whole-job units/s under weak scaling: every rank processed `units_per_rank_per_step` per step."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return units_per_rank_per_step * steps * world / elapsed
