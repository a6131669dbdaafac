"""The drop-in seams on a whole model, on the MI355X: `patch.accelerate(model)` on a ptlflow-shaped RAFT / RAFTSmall / GMA,
GPU forward vs the SAME model's unpatched CPU forward (gate: EPE <= 1e-3, BASELINE.json north_star).

Model source: the real reference (`ptlflow.models.raft.raft.RAFT` ... imported through oracle/ref_loader.py from /root/reference
or, on the GPU box, from the archive oracle/stage_ref.py staged at build time).  Without either the test is skipped, never
substituted.

Every case runs TWO consecutive forwards on different frame pairs: the reference creates `inp` / the attention map as fresh
tensors per forward and the caching allocator recycles their addresses, so a wrapper that caches by address serves pair 2
with pair 1's context features (round-1 advisor finding)."""
import pytest
import torch

from oracle import raft_oracle as O
from oracle import ref_loader

pytestmark = pytest.mark.gpu


def _run_case(model, gpu, H, W, gate=1e-3):
    from ptlflow_amd import patch
    from ptlflow_amd.update import PfkUpdateBlock
    xs = [O.smooth_pair(1, H, W, seed=11), O.smooth_pair(1, H, W, seed=12, shift=(-3, 6))]
    with torch.no_grad():
        ref = [model({"images": x.clone()})["flows"] for x in xs]           # unpatched, CPU
    patch.accelerate(model)
    try:
        assert isinstance(model.update_block, PfkUpdateBlock)
        model.to(gpu)
        with torch.no_grad():
            got = [model({"images": x.to(gpu)})["flows"].float().cpu() for x in xs]
            again = model({"images": xs[0].to(gpu)})["flows"].float().cpu()  # pair 1 after pair 2: no state leaks either way
    finally:
        patch.restore(model)
        model.cpu()
    for g, r in zip(got, ref):
        mean, mx = O.epe(g[:, 0], r[:, 0])
        assert mean <= gate, f"EPE vs the unpatched CPU forward: mean {mean:.3e} max {mx:.3e}"
    # pair 1 again after pair 2: nothing of pair 2 may survive.  Bit-identical when every op of the forward is ours; GMA's
    # stand-in attention runs torch's matmul + softmax on the GPU, whose first call may pick another kernel: allow its rounding.
    leak_mean, leak_max = O.epe(again[:, 0], got[0][:, 0])
    assert leak_mean <= 1e-4, f"pair 1 re-run after pair 2 differs: EPE mean {leak_mean:.2e} max {leak_max:.2e}"
    # the two pairs really differ (otherwise the test could not see a stale-context bug)
    assert O.epe(ref[0][:, 0], ref[1][:, 0])[0] > 0.05


# The model is the reference's own class (`ptlflow.models.raft.raft.RAFT`, `...gma.gma.GMA`), imported from /root/reference in
# the build container or from the archive `oracle/stage_ref.py` staged for the GPU box.  No stand-in: where the archive's manifest
# exists the reference MUST import (a damaged archive fails the test); on a tree with neither, the test is SKIPPED, so a GPUTEST
# record shows which model ran.
import os

REAL = ref_loader.reference_available()
_MANIFEST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "manifest.json")


def _build(kind, iters):
    torch.manual_seed(1234)
    if kind == "gma":
        return ref_loader.ref_module("ptlflow.models.gma.gma").GMA(iters=iters).eval()
    return ref_loader.build_raft(small=kind == "raft_small", iters=iters)


@pytest.mark.parametrize("kind,H,W,iters", [("raft", 436, 1024, 32), ("gma", 184, 320, 12), ("raft_small", 184, 320, 12)],
                         ids=lambda v: str(v))
def test_accelerated_model(gpu, kind, H, W, iters):
    if os.path.exists(_MANIFEST):
        assert REAL, "oracle/_ref/manifest.json is present but the reference did not import (ref_loader.REFERENCE_KIND is None)"
    if not REAL:
        pytest.skip("no reference tree and no staged archive (oracle/_ref): the reference's classes cannot run here")
    model = _build(kind, iters)
    import sys
    assert type(model).__module__ == f"ptlflow.models.{'gma.gma' if kind == 'gma' else 'raft.raft'}"
    assert sys.modules[type(model).__module__].__file__.startswith(ref_loader.REFERENCE_ROOT)
    print("model source: reference (%s)" % ref_loader.REFERENCE_KIND)
    _run_case(model, gpu, H, W)


@pytest.mark.parametrize("kind,H,W,iters", [("raft", 436, 1024, 32), ("raft", 184, 320, 5), ("gma", 184, 320, 6)], ids=lambda v: str(v))
def test_skip_dead_upsample_on_the_reference_class(gpu, kind, H, W, iters):
    """§8 f2 at the seams: `accelerate(model, skip_dead_upsample=True)` on the reference's own class returns bit-identical `flows`
    and `flow_small` (raft/raft.py:189-192) while the mask head and the upsampling run on the last iteration only — also after
    `model.iters` changes between forwards, and for a second pair (no state carried over)."""
    if not REAL:
        pytest.skip("no reference tree and no staged archive (oracle/_ref)")
    from ptlflow_amd import patch
    model = _build(kind, iters).to(gpu)
    xs = [O.smooth_pair(1, H, W, seed=11).to(gpu), O.smooth_pair(1, H, W, seed=12, shift=(-3, 6)).to(gpu)]
    patch.accelerate(model)
    try:
        with torch.no_grad():
            want = [model({"images": x}) for x in xs]
            want = [{k: v.clone() for k, v in o.items()} for o in want]
            model.iters = iters - 2
            want_short = model({"images": xs[0]})["flows"].clone()
            model.iters = iters
    finally:
        patch.restore(model)
    patch.accelerate(model, skip_dead_upsample=True)
    try:
        skip = model.update_block._skip
        assert skip is not None
        calls = {"mask": 0, "ups": 0}
        eng_conv = type(model.update_block._get_engine(gpu))._conv

        def counting_conv(self, srcs, kh, kw, key, *a, **k):
            if key == "mk":
                calls["mask"] += 1
            return eng_conv(self, srcs, kh, kw, key, *a, **k)

        seam = model.__dict__["upsample_flow"]
        kernel = seam._kernel

        def counting_kernel(flow, mask):
            calls["ups"] += 1
            return kernel(flow, mask)

        type(model.update_block._get_engine(gpu))._conv = counting_conv
        seam._kernel = counting_kernel
        try:
            with torch.no_grad():
                got = [model({"images": x}) for x in xs]
                got = [{k: v.clone() for k, v in o.items()} for o in got]
                n_mask, n_ups = calls["mask"], calls["ups"]
                model.iters = iters - 2
                got_short = model({"images": xs[0]})["flows"].clone()
                model.iters = iters
        finally:
            type(model.update_block._get_engine(gpu))._conv = eng_conv
            del seam._kernel
    finally:
        patch.restore(model)
    # (the B5 probe's own kernel call happens before counting starts only if the seam was probed earlier: allow it)
    assert n_mask == 2, f"mask conv2 ran {n_mask} times over two forwards of {iters} iterations"
    assert n_ups in (2, 3), f"convex upsampling ran {n_ups} times over two forwards"
    for g, w in zip(got, want):
        assert torch.equal(g["flows"], w["flows"]) and torch.equal(g["flow_small"], w["flow_small"])
    assert torch.equal(got_short, want_short)
    assert O.epe(want[0]["flows"][:, 0].cpu(), want[1]["flows"][:, 0].cpu())[0] > 0.05
    assert O.epe(want_short[:, 0].cpu(), want[0]["flows"][:, 0].cpu())[0] > 0
