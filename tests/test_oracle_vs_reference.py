"""Oracle and RAFT mirror vs the LIVE reference (only where /root/reference exists: the build container)."""
import pytest
import torch

from oracle import raft_oracle as O
from oracle import ref_loader

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_loader.reference_available(), reason="needs /root/reference")]


def _sd(model):
    return {k: v.detach() for k, v in model.state_dict().items() if not k.startswith("train_metrics")}


@pytest.mark.parametrize("small", [False, True])
def test_forward_bit_identical(small):
    ref = ref_loader.build_raft(small=small, iters=5)
    x = O.smooth_pair(1, 128, 192, seed=3)
    with torch.no_grad():
        r = ref({"images": x.clone()})
    o = O.raft_forward(_sd(ref), x, iters=5, small=small)
    mean, mx = O.epe(o["flows"][:, 0], r["flows"][:, 0])
    assert mean <= 1e-6 and mx <= 1e-5


@pytest.mark.parametrize("fam", ["raft", "gma", "ccmr", "ms_raft_plus"])
def test_corrblock_copies_agree(fam):
    """Every family's private CorrBlock copy computes the same thing (SURVEY finding 1) = our oracle."""
    mod = ref_loader.ref_module(f"ptlflow.models.{fam}.corr")
    g = torch.Generator().manual_seed(4)
    f1, f2 = torch.randn(1, 32, 16, 24, generator=g), torch.randn(1, 32, 16, 24, generator=g)
    cb = mod.CorrBlock(f1, f2, num_levels=4, radius=4)
    c = O.coords_grid(1, 16, 24) + torch.rand(1, 2, 16, 24, generator=g) * 12 - 6
    ref = cb(c)
    out = O.lookup(cb.corr_pyramid, c, 4)
    assert not bool(torch.isnan(ref).any()) and torch.equal(ref, out)


def test_lookup_nonfinite_matches_reference():
    mod = ref_loader.ref_module("ptlflow.models.raft.corr")
    g = torch.Generator().manual_seed(5)
    f1, f2 = torch.randn(1, 16, 8, 16, generator=g), torch.randn(1, 16, 8, 16, generator=g)
    cb = mod.CorrBlock(f1, f2, num_levels=4, radius=4)
    c = O.coords_grid(1, 8, 16)
    c[0, 0, 0, 0] = float("nan"); c[0, 1, 0, 1] = float("inf"); c[0, 0, 1, 0] = -float("inf"); c[0, 0, 1, 1] = 3e9
    ref, out = cb(c), O.lookup(cb.corr_pyramid, c, 4)
    assert bool(((ref == out) | (torch.isnan(ref) & torch.isnan(out))).all())


@pytest.mark.parametrize("small", [False, True])
def test_mirror_state_dict_is_checkpoint_compatible(small):
    from ptlflow_amd.raft import RAFT
    ref = ref_loader.build_raft(small=small)
    m = RAFT(small=small)
    sd = _sd(ref)
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd, strict=True)
    x = torch.randn(1, 3, 64, 96)
    with torch.no_grad():
        assert torch.equal(m.eval().fnet(x), ref.fnet(x)) and torch.equal(m.cnet(x), ref.cnet(x))


def test_patch_accelerate_keeps_checkpoint_keys_and_restores():
    """Seams B1/B3/B4 on the live reference model: patch, check the hooks landed where raft.py binds them,
    check state_dict keys are unchanged, check the CPU path still runs through the reference's own code
    (the hook only diverts GPU inference), restore."""
    import sys
    from ptlflow_amd import patch
    from ptlflow_amd.update import PfkUpdateBlock
    ref = ref_loader.build_raft(iters=2)
    keys = set(ref.state_dict())
    mod = sys.modules[type(ref).__module__]
    orig_fn = mod.get_corr_block
    x = O.smooth_pair(1, 128, 160, seed=5)
    with torch.no_grad():
        before = ref({"images": x.clone()})["flows"]
    try:
        patch.accelerate(ref)
    except Exception as e:  # libs not built in this checkout
        pytest.skip(f"native libs unavailable: {e}")
    from ptlflow_amd.encoder import PfkEncoder
    assert mod.get_corr_block is not orig_fn and isinstance(ref.update_block, PfkUpdateBlock)
    assert isinstance(ref.fnet, PfkEncoder) and isinstance(ref.cnet, PfkEncoder)      # seam B4
    assert set(ref.state_dict()) == keys
    xe = torch.randn(1, 3, 32, 48)
    with torch.no_grad():   # CPU tensors: the encoder wrapper defers to the reference module, list contract included
        a, b = ref.fnet([xe, xe])
        assert torch.equal(a, ref.fnet._ref[0]([xe, xe])[0]) and torch.allclose(a, b, atol=1e-5)
    # CPU tensors: corr hook falls through to the reference; update block wrapper refuses (no CPU fallback)
    cb = mod.get_corr_block(torch.randn(1, 8, 16, 16), torch.randn(1, 8, 16, 16), num_levels=4, radius=4)
    assert type(cb).__module__.startswith("ptlflow.models.raft")
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            ref({"images": x.clone()})
    patch.restore(ref)
    assert mod.get_corr_block is orig_fn and not isinstance(ref.update_block, PfkUpdateBlock)
    assert not isinstance(ref.fnet, PfkEncoder) and not isinstance(ref.cnet, PfkEncoder)
    with torch.no_grad():
        assert torch.equal(ref({"images": x.clone()})["flows"], before)


def test_alt_corr_oracle_matches_reference_iterative_block():
    """The reference's pure-torch stand-in for alt_cuda_corr (ptlflow/utils/correlation.py:539-615, selected by
    raft/corr.py:111-113 when the extension is missing) and its materialised CorrBlock both agree with the oracle's
    restatement of the CUDA kernel's arithmetic (away from the image border, where grid_sample's round trip and the
    kernel's raw floor() see the same taps)."""
    corr_mod = ref_loader.ref_module("ptlflow.models.raft.corr")
    g = torch.Generator().manual_seed(6)
    B, C, H, W = 1, 32, 16, 24
    f1, f2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    c = O.coords_grid(B, H, W) + torch.rand(B, 2, H, W, generator=g) * 6 - 3 + 0.013
    ours = O.alternate_corr_block(f1, f2, c, num_levels=3, radius=3)
    it = corr_mod.IterativeCorrBlock(fmap1=f1, fmap2=f2, radius=3, num_levels=3)(c)
    assert ours.shape == it.shape
    assert (ours - it).abs().max().item() < 1e-4
    full = corr_mod.CorrBlock(f1, f2, num_levels=3, radius=3)(c)
    assert (ours - full).abs().max().item() < 1e-4


@pytest.mark.parametrize("small", [False, True])
def test_oracle_under_autocast_is_the_reference_under_autocast(small):
    """The bf16 gate (tests/test_gpu_bf16_gate.py) measures the reference's own bf16-vs-fp32 gap by running the ORACLE under
    `torch.autocast("cpu", bfloat16)`; that is legitimate only if the oracle under autocast is the reference under autocast."""
    ref = ref_loader.build_raft(small=small, iters=6)
    x = O.smooth_pair(1, 128, 192, seed=3)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        r = ref({"images": x.clone()})["flows"].float()
        o = O.raft_forward(_sd(ref), x, iters=6, small=small)["flows"].float()
    mean, mx = O.epe(o[:, 0], r[:, 0])
    assert mean <= 1e-6 and mx <= 1e-5
    with torch.no_grad():
        gap = O.epe(r[:, 0], ref({"images": x.clone()})["flows"][:, 0])[0]
    assert gap > 1e-4        # the autocast run really is a different (bf16) computation


def test_gma_oracle_matches_reference_fp32_and_autocast():
    torch.manual_seed(3)
    m = ref_loader.ref_module("ptlflow.models.gma.gma").GMA(iters=4).eval()
    with torch.no_grad():
        m.update_block.aggregator.gamma.fill_(0.4)      # the reference initialises gamma to 0 (aggregate branch silent)
    sd = _sd(m)
    x = O.smooth_pair(1, 128, 192, seed=3)
    with torch.no_grad():
        r32, o32 = m({"images": x.clone()})["flows"], O.gma_forward(sd, x, iters=4)["flows"]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            r, o = m({"images": x.clone()})["flows"].float(), O.gma_forward(sd, x, iters=4)["flows"].float()
    assert O.epe(o32[:, 0], r32[:, 0])[1] <= 1e-5
    assert O.epe(o[:, 0], r[:, 0])[1] <= 1e-5


def _baseline_coords(B, h, w, seed):
    """SURVEY §8c's coordinate families: integer (where the grid_sample round trip flips floor() the most),
    fractional, exact halves, and far out of bounds."""
    g = torch.Generator().manual_seed(seed)
    c0 = O.coords_grid(B, h, w)
    yield "integer", c0
    yield "integer_shifted", c0 + torch.randint(-9, 10, (B, 2, h, w), generator=g).float()
    yield "fractional", c0 + torch.rand(B, 2, h, w, generator=g) * 16 - 8
    yield "half", c0 + 0.5
    yield "out_of_bounds", c0 + torch.randn(B, 2, h, w, generator=g) * 80


# 55×128 = Sintel 436×1024 (config 2), 47×156 = KITTI 375×1242 (config 4; the width with the most round-trip index flips),
# 46×62 = FlyingChairs 368×496 (config 5)
@pytest.mark.parametrize("h,w", [(55, 128), (47, 156), (46, 62)])
@pytest.mark.parametrize("D,L,r", [(256, 4, 4), (128, 2, 3)])       # D ∈ {128, 256}, L ∈ {2, 4}, r ∈ {3, 4} each appear
def test_oracle_pyramid_and_lookup_bit_exact_at_baseline_grids(h, w, D, L, r):
    """`O.correlation_pyramid` / `O.lookup` against the LIVE `raft.corr.CorrBlock` (raft/corr.py:13-64, raft/utils.py:67-81)
    at the three BASELINE grids: every pyramid level `torch.equal`, every lookup of every coordinate family `torch.equal`."""
    mod = ref_loader.ref_module("ptlflow.models.raft.corr")
    g = torch.Generator().manual_seed(1000 + h + D + 7 * L + r)
    f1, f2 = torch.randn(1, D, h, w, generator=g), torch.randn(1, D, h, w, generator=g)
    cb = mod.CorrBlock(f1, f2, num_levels=L, radius=r)
    pyr = O.correlation_pyramid(f1, f2, L)
    assert len(pyr) == len(cb.corr_pyramid) == L
    for a, b in zip(pyr, cb.corr_pyramid):
        assert torch.equal(a, b)
    for name, c in _baseline_coords(1, h, w, seed=h * w + r):
        ref = cb(c)
        out = O.lookup(pyr, c, r)
        assert ref.shape == out.shape == (1, L * (2 * r + 1) ** 2, h, w)
        assert not bool(torch.isnan(ref).any()), name
        assert torch.equal(ref, out), f"{name}: {(ref != out).sum().item()} of {ref.numel()} values differ"


@pytest.mark.reference
def test_reference_entry_points_load_unmodified():
    """`ptlflow.get_model("raft")` through the reference's own ptlflow/__init__.py + registry, and `model_benchmark.py` importable as
    a module (oracle/ref_loader.load_scripts): what tests/test_gpu_reference_scripts.py and bench.py run on the accelerated model."""
    from oracle import ref_loader
    if not ref_loader.reference_available():
        pytest.skip("no reference")
    ref_loader.load_scripts()
    import inspect
    import ptlflow
    mb = ref_loader.ref_script("model_benchmark")
    assert {"raft", "raft_small", "gma"} <= set(ptlflow.get_model_names())
    torch.manual_seed(0)
    model = ptlflow.get_model("raft_small")
    assert type(model).__name__ == "raft_small" and model.iters == 32
    src = inspect.getsource(mb.estimate_inference_time)
    assert "timer.tic()" in src and "model(inputs)" in src and mb.Timer.__module__ == "ptlflow.utils.timer"
    from jsonargparse import Namespace
    times = mb.estimate_inference_time(Namespace(num_samples=1, batch_size=1), model.eval(), (64, 128), "fp32")
    assert len(times) == 1 and times[0] > 0
