"""ptlflow_amd — MI355X-native (gfx950) kernels for the RAFT-family hot path of hmorimitsu/ptlflow.

Scope: the all-pairs correlation volume + pyramid + radius-r bilinear lookup
(`CorrBlock`) and the iterative ConvGRU/SepConvGRU update block, as hand-written HIP kernels in
``libpfk.so`` (C ABI, include/pfk.h), exposed as ``torch.ops.pfk.*`` by ``_pfk_torch.so`` and
wrapped here behind the reference's own seams:

* :func:`ptlflow_amd.corr.get_corr_block`   <-> ptlflow/models/raft/corr.py:104-118
* :class:`ptlflow_amd.update.PfkUpdateBlock` <-> ptlflow/models/raft/update.py:131-153 (`model.update_block`)
* :func:`ptlflow_amd.patch.accelerate`      monkey-patches both seams on a live ptlflow model
* :class:`ptlflow_amd.raft.RAFT`            host-side mirror of ptlflow/models/raft/raft.py:48-194
  (same state_dict names) for machines where ptlflow itself is not installed

There is no CPU / PyTorch fallback on the product path: if the native libraries are missing or
no MI355X is visible, the ops raise.
"""
from __future__ import annotations

import os
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
LIBPFK_PATH = _PKG / "libpfk.so"
TORCH_EXT_PATH = _PKG / "_pfk_torch.so"

_loaded = False


class NativeLibraryMissing(RuntimeError):
    pass


class NativeLibraryStale(RuntimeError):
    pass


def _check_stamp() -> None:
    """The libraries carry the content hash of the sources they were built from (ptlflow_amd/_build.py).  Where the sources
    lie next to them (this tree: csrc/, include/pfk.h) a library built from OTHER sources is refused — the `.so` files travel
    with the working tree and a checkout resets the mtimes a time-stamp check would rely on.  PFK_ALLOW_STALE=1 overrides."""
    if os.environ.get("PFK_ALLOW_STALE") == "1" or not (_PKG / "csrc" / "pfk_gemm.hip").exists():
        return
    from . import _build
    want = _build.source_hash()
    try:
        got = torch.ops.pfk.source_hash()          # "<extension stamp>:<libpfk.so stamp>"
    except (AttributeError, RuntimeError):         # an extension from before the stamps existed: no such op
        got = "unstamped:unstamped"
    if got != f"{want}:{want}":
        raise NativeLibraryStale(
            f"native libraries were built from other sources (stamps {got}, tree {want}); run `python -m ptlflow_amd._build` "
            "(or __graft_entry__.build()).  PFK_ALLOW_STALE=1 loads them anyway.")


def load_native(build_if_missing: bool = False) -> None:
    """Load ``_pfk_torch.so`` (which pulls in ``libpfk.so``) and register ``torch.ops.pfk``.

    Fails loudly if the libraries are not built; ``build_if_missing`` (or the environment variable
    ``PFK_AUTOBUILD=1``) compiles them in-tree first (hipcc for gfx950, a few seconds)."""
    global _loaded
    if _loaded:
        return
    autobuild = build_if_missing or os.environ.get("PFK_AUTOBUILD") == "1"
    have_sources = (_PKG / "csrc" / "pfk_gemm.hip").exists()
    if autobuild and have_sources:
        # BEFORE anything is loaded: a no-op when both libraries carry the tree's stamp, a rebuild when they are missing OR stale
        # (a stale library that is already mapped into the process cannot be replaced any more)
        from . import _build

        _build.build_all()
    if not (LIBPFK_PATH.exists() and TORCH_EXT_PATH.exists()):
        raise NativeLibraryMissing(
            f"{LIBPFK_PATH.name} / {TORCH_EXT_PATH.name} not found under {_PKG}; run "
            "`python -m ptlflow_amd._build` (or __graft_entry__.build()). There is no fallback path."
        )
    torch.ops.load_library(str(TORCH_EXT_PATH))
    _check_stamp()
    _loaded = True


def native_loaded() -> bool:
    return _loaded


def require_gpu() -> torch.device:
    load_native()
    if not torch.cuda.is_available():
        raise RuntimeError("ptlflow_amd needs a visible MI355X (torch.cuda.is_available() is False); no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


__all__ = ["load_native", "native_loaded", "require_gpu", "NativeLibraryMissing", "LIBPFK_PATH", "TORCH_EXT_PATH"]
