"""`patch.accelerate` across the reference's model zoo (CPU, build container only).

north_star's drop-in claim is about three seams of the RAFT family; everything else `ptlflow.get_model()` can return must be
left exactly as it was.  Many of the zoo's families are RAFT descendants that reuse the attribute names the seams look for
(`update_block`, `fnet` / `cnet`, `upsample_flow`, a module-level `get_corr_block`) with different shapes, arities or
semantics — csflow, lcv, llaflow, skflow, gmflownet, memflow, dip, rapidflow, rpknet, dpflow — so "accelerate is safe to call on
any model" is a property worth a test of its own: for every family that can be constructed here, the forward after
`accelerate()` (CPU tensors: no seam is eligible) equals the forward before it bit for bit, no seam wraps a block whose shapes
the kernels do not implement, and `restore()` puts the original objects back.
The hot-path families themselves (raft, gma, ccmr, ms_raft_plus, sea_raft) are covered by tests/test_patch_dispatch.py (CPU)
and tests/test_gpu_reference_models.py (MI355X)."""
import importlib
import inspect
import os
import warnings

import pytest
import torch

from oracle import ref_loader

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(ref_loader.REFERENCE_KIND not in ("tree", "env"), reason="needs the whole reference tree")]

# families that import and construct with this image's packages (the others need timm / cupy / torch_scatter / compiled
# extensions); (family, class)
ZOO = [("csflow", "CSFlow"), ("dip", "DIP"), ("dpflow", "DPFlow"), ("flow1d", "Flow1D"), ("gmflow", "GMFlow"),
       ("gmflownet", "GMFlowNet"), ("memflow", "MemFlow"),
       ("rapidflow", "RAPIDFlow"), ("rpknet", "RPKNet"), ("skflow", "SKFlow"), ("unimatch", "UniMatch"),
       ("fastflownet", "FastFlowNet"), ("hd3", "HD3"), ("irr", "IRRPWC"), ("liteflownet", "LiteFlowNet"),
       ("neuflow", "NeuFlow"), ("neuflow2", "NeuFlow2"), ("pwcnet", "PWCDCNet"), ("scopeflow", "ScopeFlow"),
       ("starflow", "StarFlow")]


def _find_class(fam, name):
    if not ref_loader.ensure_family(fam):
        pytest.skip(f"{fam}: not in this reference tree")
    root = os.path.join(ref_loader.REFERENCE_ROOT, "ptlflow", "models", fam)
    for fn in sorted(os.listdir(root)):
        if not fn.endswith(".py") or fn == "__init__.py":
            continue
        try:
            mod = importlib.import_module(f"ptlflow.models.{fam}.{fn[:-3]}")
        except Exception:       # a helper module with a dependency this image lacks
            continue
        cls = getattr(mod, name, None)
        if inspect.isclass(cls) and cls.__module__ == mod.__name__:
            return cls
    pytest.skip(f"{fam}.{name}: not importable here")


@pytest.mark.parametrize("fam,name", ZOO)
def test_accelerate_leaves_foreign_families_unchanged(fam, name):
    import ptlflow_amd
    from ptlflow_amd import patch
    from ptlflow_amd.encoder import PfkEncoder
    try:
        ptlflow_amd.load_native()
    except Exception as e:
        pytest.skip(f"native libs unavailable: {e}")
    cls = _find_class(fam, name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(0)
        model = cls().eval()
        x = {"images": torch.rand(1, 2, 3, 128, 192)}
        with torch.no_grad():
            before = model(x)["flows"]
        originals = {a: getattr(model, a, None) for a in ("update_block", "fnet", "cnet")}
        assert patch.accelerate(model) is model
        # no kernel implements these families' blocks: nothing may be wrapped (a wrapped block would refuse CPU tensors below,
        # but say it here, by name)
        assert not isinstance(getattr(model, "update_block", None), patch.PfkUpdateBlock), f"{fam}: update block wrapped"
        for a in ("fnet", "cnet"):
            enc = getattr(model, a, None)
            if isinstance(enc, PfkEncoder):          # a BasicEncoder with RAFT's own layout: allowed, and inert on CPU tensors
                assert type(enc._ref[0]).__name__ in ("BasicEncoder", "SmallEncoder")
        with torch.no_grad():
            after = model(x)["flows"]
        assert torch.equal(before, after), f"{fam}: forward changed by accelerate() (max {(before - after).abs().max():.3e})"
        patch.restore(model)
        for a, o in originals.items():
            assert getattr(model, a, None) is o, f"{fam}: restore() did not put {a} back"
        assert "upsample_flow" not in model.__dict__
        with torch.no_grad():
            assert torch.equal(model(x)["flows"], before)


@pytest.mark.parametrize("fam", ["rapidflow", "rpknet", "skflow", "matchflow"])
def test_hooked_siblings_share_rafts_corrblock(fam):
    """Seam B1 installs its hook on every model module that exposes `get_corr_block`; for families without an entry in the
    pyramid table it assumes RAFT's all-pairs block (avg-pool pyramid, radius-r window, x-offset-major).  These four zoo
    families get the hook: their private `CorrBlock` copies must BE that block — volume and lookup equal to the oracle, bit for
    bit — or the hook would change their results on the GPU."""
    from oracle import raft_oracle as O
    if not ref_loader.ensure_family(fam):
        pytest.skip(f"{fam}: not in this reference tree")
    try:
        mod = importlib.import_module(f"ptlflow.models.{fam}.corr")
    except Exception as e:
        pytest.skip(f"{fam}.corr not importable here: {e!r}")
    g = torch.Generator().manual_seed(4)
    f1, f2 = torch.randn(1, 32, 16, 24, generator=g), torch.randn(1, 32, 16, 24, generator=g)
    cb = mod.get_corr_block(f1, f2, num_levels=4, radius=4)
    assert type(cb).__name__ == "CorrBlock"
    pyr = O.correlation_pyramid(f1, f2, 4)
    for lvl, (a, b) in enumerate(zip(cb.corr_pyramid, pyr)):
        assert torch.equal(a.reshape(b.shape), b), f"{fam}: pyramid level {lvl} differs from RAFT's"
    c = O.coords_grid(1, 16, 24) + torch.rand(1, 2, 16, 24, generator=g) * 12 - 6
    assert torch.equal(cb(c), O.lookup(pyr, c, 4))


@pytest.mark.parametrize("fam,name,small", [("lcv", "LCV_RAFT", False), ("lcv", "LCV_RAFTSmall", True),
                                            ("llaflow", "LLAFlowRAFT", False), ("llaflow", "LLAFlow", False)])
def test_registered_siblings_are_wrapped(fam, name, small):
    """LCV-RAFT keeps RAFT's encoders, update block and loop (lcv/update.py and lcv/extractor.py are RAFT's files) around a
    learnable cost volume; LLA-Flow keeps RAFT's / GMA's update block and encoders around its own volume: seams B3 / B4 / B5
    apply, their own correlation code is not touched (no `get_corr_block` in those modules)."""
    import ptlflow_amd
    from ptlflow_amd import patch
    from ptlflow_amd.encoder import PfkEncoder
    try:
        ptlflow_amd.load_native()
    except Exception as e:
        pytest.skip(f"native libs unavailable: {e}")
    cls = _find_class(fam, name)
    torch.manual_seed(0)
    model = cls().eval()
    keys, corr_block = set(model.state_dict()), getattr(model, "corr_block", None)
    patch.accelerate(model)
    try:
        assert isinstance(model.update_block, patch.PfkUpdateBlock)
        assert model.update_block.spec.aggregate == (name == "LLAFlow")
        assert isinstance(model.fnet, PfkEncoder) and isinstance(model.cnet, PfkEncoder) and model.fnet.small == small
        assert getattr(model, "corr_block", None) is corr_block and set(model.state_dict()) == keys
        assert not hasattr(__import__("sys").modules[type(model).__module__], patch._ORIG)      # no correlation hook here
        if fam == "lcv":        # the learnable volume's two methods are shadowed on the module instance, the module stays
            assert patch._VOLUME_SEAM in corr_block.__dict__ and corr_block.__dict__[patch._VOLUME_SEAM].channels_last is True
        with pytest.raises(RuntimeError, match="GPU tensors"):       # no CPU fallback behind a wrapped block
            with torch.no_grad():
                model({"images": torch.rand(1, 2, 3, 128, 192)})
    finally:
        patch.restore(model)
    assert not isinstance(model.update_block, patch.PfkUpdateBlock) and not isinstance(model.fnet, PfkEncoder)
    if corr_block is not None:
        assert not {patch._VOLUME_SEAM, "forward", "compute_cost_volume"} & set(corr_block.__dict__)


def test_lcv_volume_is_rafts_block_on_a_transformed_map():
    """What `patch._LearnableVolumeSeam` relies on: lcv/corr_lcv.py's volume `(fmap1' W) fmap2 / sqrt(D)`, pyramid and lookup equal
    RAFT's block (the oracle) built from the W-transformed first feature map — bit for bit on the CPU, for a W that is not the
    identity, at a size where the module pools three times; and the module stops pooling (its levels repeat) below that size,
    which is where the seam must leave it alone."""
    from oracle import raft_oracle as O
    from ptlflow_amd import patch
    if not ref_loader.ensure_family("lcv"):
        pytest.skip("lcv: not in this reference tree")
    mod = importlib.import_module("ptlflow.models.lcv.corr_lcv")
    torch.manual_seed(1)
    cb = mod.LearnableCorrBlock(64, 4, 4)
    with torch.no_grad():
        cb.raw_P.add_(torch.randn(64, 64) * 0.3)
        cb.raw_D.add_(torch.randn(64) * 0.5)
        f1, f2 = torch.randn(2, 64, 44, 48), torch.randn(2, 64, 44, 48)
        pyr = cb.compute_cost_volume(f1, f2)
        f1w = torch.matmul(f1.flatten(2).transpose(1, 2), cb.W).view(2, 44, 48, 64).permute(0, 3, 1, 2)
        want = O.correlation_pyramid(f1w, f2, 4)
        for a, b in zip(pyr, want):
            assert torch.equal(a.reshape(b.shape), b)
        c = O.coords_grid(2, 44, 48) + torch.rand(2, 2, 44, 48) * 6 - 3
        assert torch.equal(cb(pyr, c), O.lookup(want, c, 4))
        small = cb.compute_cost_volume(f1[..., :32, :40].contiguous(), f2[..., :32, :40].contiguous())
    assert [tuple(p.shape[-2:]) for p in small[:4]] == [(32, 40), (16, 20), (8, 10), (8, 10)]      # no third pooling
    seam = patch._LearnableVolumeSeam(cb)
    assert not seam.eligible(f1, f2)                      # CPU tensors
    assert any(min(32 >> i, 40 >> i) <= 9 for i in range(3))       # the size rule of `eligible` for the case above
