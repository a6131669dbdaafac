"""Seam B1 inside the reference's sibling families — the real `RAPIDFlow`, `RPKNet` and `SKFlow` classes on the MI355X.

Their model modules import a `get_corr_block` whose `CorrBlock` is RAFT's, copy for copy (tests/test_patch_zoo.py pins each
against the oracle, bit for bit), so `patch.accelerate` installs the correlation hook in them as well — with shapes the RAFT
tests never produce: rapidflow / rpknet build ONE-level pyramids (num_levels = 1, radius 4) on 128-channel maps at three
scales down to 4x6 pixels (rapidflow.py:300, rpknet.py), skflow the four-level one on 256 channels.  Everything else in these
models (NeXt1D / PKConv encoders, their own update blocks) is foreign to the kernels and must stay the reference's torch code.

Checked: the hook served every correlation block of the forward; the accelerated forward equals the SAME model's stock
PyTorch-ROCm forward on the GPU and its CPU forward to fp32 re-association noise; `restore()` gives the stock result back.
The classes come from /root/reference or, on the GPU box, from the archive oracle/stage_ref.py staged at build time."""
import copy
import sys
import warnings

import pytest
import torch

from oracle import raft_oracle as O
from oracle import ref_loader

pytestmark = [pytest.mark.gpu, pytest.mark.reference,
              pytest.mark.skipif(not ref_loader.reference_available(),
                                 reason="no reference: run `python -c 'import __graft_entry__ as g; g.build()'` where "
                                        "/root/reference exists; it stages oracle/_ref/ for the GPU box")]

# family, module, class, correlation blocks per forward at 128x192 (rapidflow / rpknet: one per pyramid scale)
SIBLINGS = [("rapidflow", "rapidflow", "RAPIDFlow", 3), ("rpknet", "rpknet", "RPKNet", 3), ("skflow", "skflow", "SKFlow", 1)]


@pytest.mark.parametrize("fam,modname,clsname,blocks", SIBLINGS)
def test_sibling_family_whole_model(gpu, fam, modname, clsname, blocks):
    from ptlflow_amd import patch
    assert ref_loader.ensure_family(fam), f"{fam} was not staged"
    mod = ref_loader.ref_module(f"ptlflow.models.{fam}.{modname}")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(1234)
        model = getattr(mod, clsname)().eval()
    x = O.smooth_pair(1, 128, 192, seed=11)
    with torch.no_grad():
        # the CPU forward runs on a copy: rapidflow / rpknet cache derived weights in plain attributes during a forward
        # (next1d.py:135, pkconv.py:95), which `.to(device)` does not move
        cpu = copy.deepcopy(model)({"images": x.clone()})["flows"][:, 0]
        model.to(gpu)
        stock = model({"images": x.to(gpu)})["flows"][:, 0].float().cpu()
        stock2 = model({"images": x.to(gpu)})["flows"][:, 0].float().cpu()    # the stock forward's own run-to-run spread
    update_block, fnet = model.update_block, getattr(model, "fnet", None)
    hook_before = mod.get_corr_block
    patch.accelerate(model)
    assert mod.get_corr_block is not hook_before
    served = [0]
    orig = patch._pfk_get_corr_block

    def counting(*a, **k):
        served[0] += 1
        block = orig(*a, **k)
        assert block.channels_last is False, "a torch consumer must get plain NCHW lookups (memory-format propagation)"
        return block

    patch._pfk_get_corr_block = counting
    try:
        assert model.update_block is update_block and getattr(model, "fnet", None) is fnet, "a foreign block was wrapped"
        with torch.no_grad():
            got = model({"images": x.to(gpu)})["flows"][:, 0].float().cpu()
        assert served[0] == blocks, f"seam B1 served {served[0]} correlation blocks of {clsname}.forward, expected {blocks}"
    finally:
        patch._pfk_get_corr_block = orig
        patch.restore(model)
    with torch.no_grad():
        again = model({"images": x.to(gpu)})["flows"][:, 0].float().cpu()
    assert torch.isfinite(got).all()
    scale = float(cpu.abs().max()) + 1e-6
    e_stock = O.epe(got, stock)
    e_cpu = O.epe(got, cpu)
    e_base = O.epe(stock, cpu)       # what stock GPU kernels (rocBLAS / MIOpen) already differ from the CPU by
    print(f"{clsname}: |flow| max {scale:.3f}; accelerated vs stock GPU EPE mean {e_stock[0]:.3e} max {e_stock[1]:.3e}; vs CPU "
          f"{e_cpu[0]:.3e} / {e_cpu[1]:.3e}; stock GPU vs CPU {e_base[0]:.3e} / {e_base[1]:.3e}", file=sys.stderr)
    # random-init networks, a dozen refinement steps: fp32 re-association noise stays at the 1e-5 level relative to the flow
    assert e_stock[0] <= 2e-5 * max(1.0, scale) and e_stock[1] <= 2e-4 * max(1.0, scale)
    assert e_cpu[0] <= 2e-5 * max(1.0, scale) + 2 * e_base[0]
    # restore(): the module's own function object is back, no instance attribute shadows the class's methods, and the forward
    # is the stock one again (bit for bit where the stock forward is run-to-run deterministic; within its own spread where
    # MIOpen's kernels are not)
    assert mod.get_corr_block is hook_before and "upsample_flow" not in model.__dict__
    spread = O.epe(stock2, stock)[1]
    print(f"{clsname}: stock run-to-run max {spread:.3e}; after restore() vs stock max {O.epe(again, stock)[1]:.3e}", file=sys.stderr)
    if spread == 0.0:
        assert torch.equal(again, stock), "restore() did not give the stock forward back"
    else:       # two samples of a run-to-run spread: leave room for a third one to land further out
        assert O.epe(again, stock)[1] <= max(4 * spread, 2e-5 * max(1.0, scale))


@pytest.mark.parametrize("fam,modname,clsname,small", [("lcv", "lcv_raft", "LCV_RAFT", False), ("lcv", "lcv_raft", "LCV_RAFTSmall", True),
                                                      ("llaflow", "llaflow", "LLAFlowRAFT", False), ("llaflow", "llaflow", "LLAFlow", False)])
def test_registered_sibling_whole_model(gpu, fam, modname, clsname, small):
    """LCV-RAFT (lcv/lcv_raft.py:124-189): RAFT's encoders, update block and loop around a learnable cost volume — lcv/update.py
    and lcv/extractor.py are RAFT's files.  LLA-Flow (llaflow/llaflow.py:150-215): RAFT's block (`LLAFlowRAFT`) or GMA's with one
    head (`LLAFlow`, attention passed as the fifth argument) and RAFT's encoders around its own cost volume.  Seams B3 (update
    block), B4 (encoders) and B5 (`upsample_flow`) serve the real classes; lcv's learnable volume `(fmap1' W) fmap2` additionally
    goes through B1 (`patch._LearnableVolumeSeam`: K1-K3 on the W-transformed feature map), llaflow's stays the reference's."""
    from ptlflow_amd import patch
    from ptlflow_amd.encoder import PfkEncoder
    assert ref_loader.ensure_family(fam), f"{fam} was not staged"
    mod = ref_loader.ref_module(f"ptlflow.models.{fam}.{modname}")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(1234)
        model = getattr(mod, clsname)().eval()
    # 320 x 448: 40 x 56 maps, the smallest at which lcv's pyramid still pools three times (corr_lcv.py:46-49) — below that its
    # levels repeat and the learnable-volume seam leaves the module alone
    x = O.smooth_pair(1, 320, 448, seed=12)
    if fam == "lcv":        # a W that is not the identity it is initialised to (raw_P = I, raw_D = 0)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            model.corr_block.raw_P.add_(torch.randn(model.corr_block.raw_P.shape, generator=g) * 0.05)
            model.corr_block.raw_D.add_(torch.randn(model.corr_block.raw_D.shape, generator=g) * 0.3)
    with torch.no_grad():
        cpu = copy.deepcopy(model)({"images": x.clone()})["flows"][:, 0]
        model.to(gpu)
        stock = model({"images": x.to(gpu)})["flows"][:, 0].float().cpu()
    patch.accelerate(model)
    served = [0]
    orig_block = patch._pfk_get_corr_block

    def counting(*a, **k):
        served[0] += 1
        return orig_block(*a, **k)

    patch._pfk_get_corr_block = counting
    try:
        assert isinstance(model.update_block, patch.PfkUpdateBlock) and isinstance(model.fnet, PfkEncoder)
        assert isinstance(model.cnet, PfkEncoder) and model.fnet.small == small
        with torch.no_grad():
            got = model({"images": x.to(gpu)})["flows"][:, 0].float().cpu()
        assert small or model.__dict__["upsample_flow"].ok is True, f"seam B5 rejected {clsname}.upsample_flow"
        # lcv: the learnable volume is served by K1-K3 (one block per forward); llaflow builds its own volume directly
        assert served[0] == (1 if fam == "lcv" else 0)
    finally:
        patch._pfk_get_corr_block = orig_block
        patch.restore(model)
    assert "forward" not in getattr(model, "corr_block", model).__dict__
    assert torch.isfinite(got).all()
    scale = float(cpu.abs().max()) + 1e-6
    e_stock, e_cpu, e_base = O.epe(got, stock), O.epe(got, cpu), O.epe(stock, cpu)
    print(f"{clsname}: |flow| max {scale:.3f}; accelerated vs stock GPU EPE mean {e_stock[0]:.3e} max {e_stock[1]:.3e}; vs CPU "
          f"{e_cpu[0]:.3e} / {e_cpu[1]:.3e}; stock GPU vs CPU {e_base[0]:.3e} / {e_base[1]:.3e}", file=sys.stderr)
    assert e_cpu[0] <= 2e-5 * max(1.0, scale) + 2 * e_base[0] and e_cpu[1] <= 2e-4 * max(1.0, scale) + 2 * e_base[1]
    assert e_stock[0] <= 2e-5 * max(1.0, scale) + 2 * e_base[0]
