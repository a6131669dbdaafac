"""The reference's OWN model classes on the MI355X with the seams installed — whole models of the families §8 lists beside
RAFT / GMA (tests/test_gpu_live_model.py has those two):

* `ptlflow.models.sea_raft.sea_raft.SEARAFT` (sea_raft.py:209-224): seam B1 inside the real model — its ResNet-FPN encoders and
  ConvNeXt update block stay the reference's torch code (MIOpen), `get_corr_block` is served by K1-K3 with the bilinear-½
  pyramid; fp32 vs the model's own CPU forward, and under bf16 autocast vs the model's CPU autocast gap;
* `ptlflow.models.ccmr.ccmr.CCMR` and `ptlflow.models.ms_raft_plus.ms_raft_plus.MSRAFTPlus` constructed with their DEFAULTS —
  `alternate_corr=True` (ccmr.py:52, ms_raft_plus.py:78) — so that the reference's own `AlternateCorrBlock` (ccmr/corr.py:68-101)
  calls the module it imported as `alt_cuda_corr`: this repo's plug-in (seam B2), zero patching; `accelerate` adds B3 (the
  update block) on top.  The CPU side of the comparison is the same object with `alternate_corr=False` (the reference has no
  CPU implementation of the extension);
* `RAFT.training_step` (base_model.py:322-361) with the reference's `SequenceLoss` (raft/raft.py:20-45) driving the libpfk
  autograd nodes through the seams, every parameter's gradient vs the same model's float64 CPU autograd.

The classes are imported by oracle/ref_loader.py from /root/reference, or — on the GPU box — from the archive
oracle/stage_ref.py staged at build time (the same unmodified files).  Without either there is nothing to test here."""
import sys

import pytest
import torch

from oracle import raft_oracle as O
from oracle import ref_loader

pytestmark = [pytest.mark.gpu, pytest.mark.reference,
              pytest.mark.skipif(not ref_loader.reference_available(),
                                 reason="no reference: run `python -c 'import __graft_entry__ as g; g.build()'` where "
                                        "/root/reference exists; it stages oracle/_ref/ for the GPU box")]


def _counting(obj, name):
    """Wrap obj.name with a call counter; returns (counter list, undo)."""
    orig = getattr(obj, name)
    n = [0]

    def wrapped(*a, **k):
        n[0] += 1
        return orig(*a, **k)

    setattr(obj, name, wrapped)
    return n, lambda: setattr(obj, name, orig)


def test_sea_raft_whole_model(gpu):
    from ptlflow_amd import patch
    S = ref_loader.ref_module("ptlflow.models.sea_raft.sea_raft")
    torch.manual_seed(1234)
    # (a list: the tuple default is mutated in place by sea_raft/extractor.py:32-33 when not routed through jsonargparse;
    #  `pretrain="resnet18"` only selects the block counts here — init_weight=False, nothing is downloaded)
    model = S.SEARAFT(block_dims=[64, 128, 256]).eval()
    H, W = 436, 1024
    x = O.smooth_pair(1, H, W, seed=11)
    with torch.no_grad():
        ref = model({"images": x.clone()})["flows"]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ref_bf = model({"images": x.clone()})["flows"].float()
    gap = O.epe(ref_bf[:, 0], ref[:, 0])[0]
    # the same model under bf16 autocast on STOCK PyTorch-ROCm ops (un-patched): autocast casts other ops on the GPU than on the
    # CPU (and MIOpen's bf16 convolutions are not oneDNN's), so this — not the CPU autocast run — is what the seam must match
    model.to(gpu)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        stock_bf = model({"images": x.to(gpu)})["flows"].float().cpu()
    model.cpu()
    gap_stock = O.epe(stock_bf[:, 0], ref[:, 0])[0]
    patch.accelerate(model)
    calls, undo = _counting(patch, "_pfk_get_corr_block")
    try:
        # SEA-RAFT's update block / encoders are other layer types: they must be left alone (dispatch by implementation)
        assert type(model.update_block).__module__ == "ptlflow.models.sea_raft.update"
        model.to(gpu)
        with torch.no_grad():
            got = model({"images": x.to(gpu)})["flows"].float().cpu()
            assert calls[0] == 1, "seam B1 was not used by SEARAFT.forward"
            with torch.autocast("cuda", dtype=torch.bfloat16):
                got_bf = model({"images": x.to(gpu)})["flows"].float().cpu()
            assert calls[0] == 2
    finally:
        undo()
        patch.restore(model)
        model.cpu()
    mean, mx = O.epe(got[:, 0], ref[:, 0])
    mean_bf, mx_bf = O.epe(got_bf[:, 0], ref[:, 0])
    seam_vs_stock = O.epe(got_bf[:, 0], stock_bf[:, 0])[0]
    print(f"SEARAFT 436x1024 fp32: EPE vs its own CPU forward mean {mean:.3e} max {mx:.3e}; bf16 autocast vs CPU fp32: seam "
          f"{mean_bf:.3e}, stock PyTorch-ROCm autocast {gap_stock:.3e}, CPU autocast {gap:.3e}; seam vs stock GPU autocast {seam_vs_stock:.3e}")
    assert mean <= 1e-3
    # bf16: no further from the fp32 forward than twice what bf16 autocast costs this model WITHOUT the seam (GPU or CPU)
    assert mean_bf <= 2.0 * max(gap, gap_stock) + 1e-3


@pytest.mark.parametrize("family,H,W", [("ccmr", 256, 384), ("ms_raft_plus", 192, 256)])
def test_alternate_corr_families_with_defaults(gpu, family, H, W):
    import alt_cuda_corr                                  # the repo-root plug-in: what `import alt_cuda_corr` resolves to
    from ptlflow_amd import patch
    from ptlflow_amd.update import PfkUpdateBlock
    assert alt_cuda_corr.forward.__module__ == "ptlflow_amd.altcorr"
    if family == "ccmr":
        M = ref_loader.ref_module("ptlflow.models.ccmr.ccmr")
        torch.manual_seed(1234)
        model = M.CCMR().eval()                            # defaults: alternate_corr=True, iters [8, 10, 15]
    else:
        M = ref_loader.ref_module("ptlflow.models.ms_raft_plus.ms_raft_plus")
        torch.manual_seed(1234)
        model = M.MSRAFTPlus().eval()                      # defaults: alternate_corr=True, iters (4, 6, 5, 10)
    assert model.alternate_corr is True
    corr_mod = sys.modules[f"ptlflow.models.{family}.corr"]
    assert corr_mod.alt_cuda_corr is alt_cuda_corr, "the reference's corr.py did not pick the plug-in up"
    x = O.smooth_pair(1, H, W, seed=11)
    model.alternate_corr = False                           # CPU side: the materialised CorrBlock (no CPU extension exists)
    with torch.no_grad():
        ref = model({"images": x.clone()})["flows"]
    model.alternate_corr = True
    fwd_calls, undo = _counting(alt_cuda_corr, "forward")
    patch.accelerate(model)
    try:
        assert isinstance(model.update_block, PfkUpdateBlock), "seam B3 did not match the family's update block"
        model.to(gpu)
        with torch.no_grad():
            got = model({"images": x.to(gpu)})["flows"].float().cpu()
            n_first = fwd_calls[0]
            again = model({"images": x.to(gpu)})["flows"].float().cpu()
    finally:
        undo()
        patch.restore(model)
        model.cpu()
    assert n_first > 0, "alt_cuda_corr.forward was never called: the on-demand kernel did not run"
    mean, mx = O.epe(got[:, 0], ref[:, 0])
    print(f"{family} {H}x{W} (defaults, alternate_corr=True, {n_first} alt_cuda_corr.forward calls): EPE vs the CPU forward "
          f"(alternate_corr=False) mean {mean:.3e} max {mx:.3e}")
    assert torch.isfinite(got).all()
    assert mean <= 1e-3
    assert O.epe(again[:, 0], got[:, 0])[0] <= 1e-5        # a second forward on the same pair: no state carried over


def test_reference_training_step(gpu):
    """`RAFT.training_step` of the reference (forward in train mode, its SequenceLoss, its FlowMetrics) on the accelerated
    model: loss and every parameter's gradient vs float64 autograd of the SAME class on the CPU; the fp32 CPU run of the
    same class says how much of the difference is fp32 conditioning."""
    from ptlflow_amd import patch
    from tests.test_gpu_train_step import compare_gradients
    B, H, W, iters = 2, 368, 496, 3
    g = torch.Generator().manual_seed(9)
    batch = {"images": torch.rand(B, 2, 3, H, W, generator=g), "flows": torch.randn(B, 1, 2, H, W, generator=g) * 4,
             "valids": (torch.rand(B, 1, 1, H, W, generator=g) > 0.1).float(), "meta": {"dataset_name": "synthetic"}}

    def cpu_run(dtype):
        m = ref_loader.build_raft(iters=iters, seed=21).to(dtype).train()
        b = {k: (v.to(dtype) if torch.is_tensor(v) else v) for k, v in batch.items()}
        loss = m.training_step(b, 0)["loss"]
        loss.backward()
        return loss.detach(), {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}

    loss64, g64 = cpu_run(torch.float64)
    _, g32 = cpu_run(torch.float32)
    model = ref_loader.build_raft(iters=iters, seed=21).train()
    patch.accelerate(model)
    try:
        model.to(gpu)
        out = model.training_step({k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in batch.items()}, 0)
        out["loss"].backward()
        got = {n: (None if p.grad is None else p.grad.detach().double().cpu()) for n, p in model.named_parameters()}
    finally:
        patch.restore(model)
    assert abs(out["loss"].item() - loss64.item()) <= 1e-4 * abs(loss64.item())
    compare_gradients(got, g64, g32, tol=5e-4, elem_mult=15.0, l2_mult=5.0, iters=iters)
