"""`tests/seam_model.py::SeamRAFT` — the torch-only, reference-shaped caller of seams B1/B3/B4 used by bench.py's `dropin`
leg — computes what the reference computes (CPU, against the oracle), carries the reference's state_dict layout, and is
recognised by `patch.accelerate` exactly like a ptlflow model (dispatch by module + class + shapes; restore undoes it)."""
import sys

import pytest
import torch

from oracle import raft_oracle as O


def _pair(small, iters, seed=5):
    from ptlflow_amd.raft import RAFT
    from seam_model import SeamRAFT
    mirror = RAFT(small=small, iters=iters).load_synthetic(seed).eval()
    seam = SeamRAFT(small=small, iters=iters).eval()
    missing = seam.load_state_dict(mirror.state_dict(), strict=True)       # same keys, same shapes as the mirror (= the reference's)
    assert not missing.missing_keys and not missing.unexpected_keys
    return mirror, seam


@pytest.mark.parametrize("small,H,W,iters", [(False, 128, 192, 4), (True, 128, 200, 3), (False, 131, 171, 2)])
def test_unpatched_seam_model_matches_oracle_on_cpu(small, H, W, iters):
    mirror, seam = _pair(small, iters)
    x = O.smooth_pair(2, H, W, seed=3)
    ref = O.raft_forward({k: v.clone() for k, v in mirror.state_dict().items()}, x, iters=iters, small=small)
    out = seam({"images": x})
    assert out["flows"].shape == ref["flows"].shape == (2, 1, 2, H, W)
    mean, mx = O.epe(out["flows"][:, 0], ref["flows"][:, 0])
    assert mean <= 1e-5 and mx <= 1e-4, f"EPE vs oracle: mean {mean:.2e} max {mx:.2e}"
    ms, _ = O.epe(out["flow_small"], ref["flow_small"])
    assert ms <= 1e-5


def test_torch_corr_block_matches_oracle_lookup():
    from seam_model import TorchCorrBlock
    g = torch.Generator().manual_seed(1)
    f1, f2 = torch.randn(2, 64, 17, 21, generator=g), torch.randn(2, 64, 17, 21, generator=g)
    coords = O.coords_grid(2, 17, 21) + torch.rand(2, 2, 17, 21, generator=g) * 12 - 6
    got = TorchCorrBlock(f1, f2, 4, 4)(coords)
    want = O.lookup(O.correlation_pyramid(f1, f2, 4), coords, 4)
    assert got.shape == want.shape == (2, 324, 17, 21)
    assert float((got - want).abs().max()) <= 1e-5


@pytest.mark.parametrize("small", [False, True])
def test_accelerate_dispatches_on_seam_model(small):
    import seam_model
    from ptlflow_amd import patch
    from ptlflow_amd.encoder import PfkEncoder
    from ptlflow_amd.update import PfkUpdateBlock
    _, seam = _pair(small, 2)
    keys = list(seam.state_dict().keys())
    orig = seam_model.get_corr_block
    try:
        patch.accelerate(seam)
    except RuntimeError as e:                      # no libpfk.so in this checkout: nothing to dispatch to
        pytest.skip(str(e))
    try:
        assert isinstance(seam.update_block, PfkUpdateBlock)
        assert isinstance(seam.fnet, PfkEncoder) and isinstance(seam.cnet, PfkEncoder)
        assert sys.modules[type(seam).__module__].get_corr_block is not orig          # seam B1 rebound on THIS module
        assert list(seam.state_dict().keys()) == keys                                  # checkpoint layout untouched
        # CPU tensors fall through the hook to the module's own torch block
        f = torch.randn(1, 32, 8, 8)
        assert isinstance(seam_model.get_corr_block(f, f), seam_model.TorchCorrBlock)
    finally:
        patch.restore(seam)
    assert seam_model.get_corr_block is orig and not isinstance(seam.update_block, PfkUpdateBlock)
