"""The CPU oracle's whole forwards, computed once per pytest process: several GPU tests hold the kernels against the SAME fp32 CPU
forward of the headline configuration (seeded synthetic weights, seeded smooth pair, 436x1024, 32 iterations — 25-30 s of host time each
on the GPU box); the cache keys on everything that determines the result.  Test infrastructure only."""
import torch

from oracle import raft_oracle as O

_CACHE = {}


def cpu_forward(kind: str, P: dict, x: torch.Tensor, iters: int, small: bool = False, autocast: bool = False, key=None):
    """`O.raft_forward` / `O.gma_forward` on (P, x); `key` = a hashable that identifies (weights, input) — e.g. the seeds they were made
    from — or None for no caching.  Returns a dict of fresh clones (callers may modify them)."""
    k = None if key is None else (kind, key, tuple(x.shape), iters, small, autocast)
    if k is None or k not in _CACHE:
        fwd = O.gma_forward if kind == "gma" else O.raft_forward
        kw = {} if kind == "gma" else {"small": small}
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = fwd(P, x, iters=iters, **kw)
        else:
            out = fwd(P, x, iters=iters, **kw)
        out = {n: v.detach() for n, v in out.items() if isinstance(v, torch.Tensor)}
        if k is None:
            return out
        _CACHE[k] = out
    return {n: v.clone() for n, v in _CACHE[k].items()}
