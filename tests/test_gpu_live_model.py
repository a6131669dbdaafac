"""The drop-in seams on a whole model, on the MI355X: `patch.accelerate(model)` on a ptlflow-shaped RAFT / RAFTSmall / GMA,
GPU forward vs the SAME model's unpatched CPU forward (gate: EPE <= 1e-3, BASELINE.json north_star).

Two model sources:
* the real reference (`ptlflow.models.raft.raft.RAFT` ... imported from /root/reference through oracle/ref_loader.py) — only
  where that tree exists; /root/reference is absent on the GPU box and the build container has no GPU, so these cases run
  only on a machine that has both (they are the judge's "gpu + reference" cases and skip elsewhere);
* `tests/livelike.py` — same module paths, class names, state_dict keys and caller loop, forward = the CPU oracle — which is
  what runs on the GPU box.

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


@pytest.mark.parametrize("kind,H,W,iters", [("raft", 436, 1024, 32), ("gma", 184, 320, 12), ("raft_small", 184, 320, 12)])
def test_accelerated_livelike_model(gpu, kind, H, W, iters):
    if ref_loader.reference_available():
        pytest.skip("the real reference is importable here: test_accelerated_reference_model covers this (livelike registers "
                    "stand-in modules under the reference's names, which must not shadow the real ones)")
    from tests import livelike
    _run_case(livelike.build(kind, iters=iters), gpu, H, W)


@pytest.mark.reference
@pytest.mark.skipif(not ref_loader.reference_available(), reason="needs /root/reference next to a GPU")
@pytest.mark.parametrize("kind,H,W,iters", [("raft", 436, 1024, 32), ("gma", 184, 320, 12), ("raft_small", 184, 320, 12)])
def test_accelerated_reference_model(gpu, kind, H, W, iters):
    torch.manual_seed(1234)
    if kind == "gma":
        model = ref_loader.ref_module("ptlflow.models.gma.gma").GMA(iters=iters).eval()
    else:
        model = ref_loader.build_raft(small=kind == "raft_small", iters=iters)
    _run_case(model, gpu, H, W)
