"""The drop-in path a ptlflow user gets, on the MI355X: `tests/seam_model.py::SeamRAFT` — a torch-only RAFT whose forward IS
the reference's caller loop (raft.py:125-194: module-global `get_corr_block`, `update_block(net, inp, corr, flow)` on NCHW
tensors, `upsample_flow` in torch ops) — with `patch.accelerate` applied, against the CPU oracle (gate: EPE <= 1e-3).
bench.py's `dropin` leg times exactly this object."""
import pytest
import torch

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


def _models(small, iters, seed):
    from ptlflow_amd.raft import RAFT
    from seam_model import SeamRAFT
    mirror = RAFT(small=small, iters=iters).load_synthetic(seed).eval()
    P = {k: v.clone() for k, v in mirror.state_dict().items()}
    seam = SeamRAFT(small=small, iters=iters).eval()
    seam.load_state_dict(P, strict=True)
    return P, seam


@pytest.mark.parametrize("small,H,W,iters", [(False, 436, 1024, 6), (True, 184, 320, 12), (False, 375, 1242, 4)])
def test_accelerated_seam_model_vs_oracle(gpu, small, H, W, iters):
    import seam_model
    from ptlflow_amd import patch
    from ptlflow_amd.corr import CorrBlock
    from ptlflow_amd.encoder import PfkEncoder
    from ptlflow_amd.update import PfkUpdateBlock
    P, seam = _models(small, iters, 1234)
    xs = [O.smooth_pair(1, H, W, seed=11), O.smooth_pair(1, H, W, seed=12, shift=(-3, 6))]
    refs = [O.raft_forward(P, x, iters=iters, small=small) for x in xs]
    seam = seam.to(gpu)
    patch.accelerate(seam)
    try:
        assert isinstance(seam.update_block, PfkUpdateBlock) and isinstance(seam.fnet, PfkEncoder)
        f = torch.randn(1, 64, 16, 24, device=gpu)
        assert isinstance(seam_model.get_corr_block(f, f), CorrBlock)                 # seam B1 is live on this module
        outs = [seam({"images": x.to(gpu)}) for x in xs]
        again = seam({"images": xs[0].to(gpu)})
        if not small:     # seam B5: the model's own upsample_flow method is shadowed by the kernel after the behaviour probe
            assert isinstance(seam.upsample_flow, patch._UpsampleSeam) and seam.upsample_flow.ok is True
    finally:
        patch.restore(seam)
    for o, r in zip(outs, refs):
        assert tuple(o["flows"].shape) == (1, 1, 2, H, W)
        mean, mx = O.epe(o["flows"][:, 0].float().cpu(), r["flows"][:, 0])
        print(f"seam path {H}x{W} {iters} it: EPE vs CPU oracle mean {mean:.3e} max {mx:.3e}")
        assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"
        ms, _ = O.epe(o["flow_small"].float().cpu(), r["flow_small"])
        assert ms <= 1e-3
    # pair 1 again after pair 2: every op on this path is libpfk's or an elementwise torch op -> bit-identical
    assert torch.equal(again["flows"], outs[0]["flows"])


def test_upsample_seam_dispatches_by_behaviour(gpu):
    """Seam B5 replaces `model.upsample_flow` only if the model's method agrees with the kernel on a probe: a method that computes
    something else (here: a x4 factor) keeps running its own code; gradient-carrying and CPU calls always do."""
    from ptlflow_amd import patch
    from seam_model import SeamRAFT, convex_upsample_torch

    class Odd(SeamRAFT):
        def upsample_flow(self, flow, mask):
            return 0.5 * convex_upsample_torch(flow, mask)

    g = torch.Generator().manual_seed(2)
    flow, mask = torch.randn(1, 2, 6, 9, generator=g).to(gpu), torch.randn(1, 576, 6, 9, generator=g).to(gpu)
    for cls, expect in ((SeamRAFT, True), (Odd, False)):
        m = cls(iters=1).eval().to(gpu)
        want = m.upsample_flow(flow, mask)
        patch.accelerate(m)
        try:
            got = m.upsample_flow(flow, mask)
            assert m.upsample_flow.ok is expect
            assert float((got - want).abs().max()) <= 1e-5
            # a gradient graph goes to the original
            fg = flow.clone().requires_grad_(True)
            assert m.upsample_flow(fg, mask).requires_grad
        finally:
            patch.restore(m)
        assert "upsample_flow" not in m.__dict__


def test_seam_model_unpatched_gpu_is_the_torch_path(gpu):
    """Un-patched, the same object runs on stock PyTorch-ROCm ops (MIOpen convolutions, grid_sample): the baseline a ptlflow
    user has on this GPU today.  Looser gate: MIOpen's algorithms are not the oracle's."""
    P, seam = _models(False, 4, 7)
    x = O.smooth_pair(1, 184, 320, seed=3)
    ref = O.raft_forward(P, x, iters=4)
    out = seam.to(gpu)({"images": x.to(gpu)})
    mean, mx = O.epe(out["flows"][:, 0].float().cpu(), ref["flows"][:, 0])
    assert mean <= 1e-3, f"EPE mean {mean:.2e} max {mx:.2e}"


def test_fp16_feature_maps_keep_an_fp32_volume(gpu):
    """`model.half()` is the reference's reduced-precision mode (validate.py:243-244): families whose encoder is not wrapped
    (sea_raft, ccmr, ms_raft_plus) hand fp16 maps to the B1 hook.  fp16 operands are exact in fp32, so the block must keep the
    fp32 volume (not round the maps to bf16) and return fp16 like the reference's `corr.to(coords.dtype)`-style cast."""
    import seam_model
    from ptlflow_amd import patch
    g = torch.Generator().manual_seed(5)
    f1 = (torch.randn(2, 128, 20, 28, generator=g) * 0.5).half()
    f2 = (torch.randn(2, 128, 20, 28, generator=g) * 0.5).half()
    coords = O.coords_grid(2, 20, 28) + torch.rand(2, 2, 20, 28, generator=g) * 10 - 5
    hook = patch._make_corr_hook(seam_model.__name__, seam_model.get_corr_block)
    blk = hook(fmap1=f1.to(gpu), fmap2=f2.to(gpu), num_levels=4, radius=4)
    assert blk.volume_dtype == torch.float32 and blk.corr_pyramid[0].dtype == torch.float32
    out = blk(coords.to(gpu))
    assert out.dtype == torch.float16
    want = O.lookup(O.correlation_pyramid(f1.float(), f2.float(), 4), coords, 4)
    # the fp32 lookup itself (before the final cast) is the fp32 path's: compare at fp16 resolution of the reference values
    err = (out.float().cpu() - want).abs()
    assert float((err - want.abs() * 2.0 ** -10).clamp_min(0).max()) <= 2e-3, float(err.max())
    # bf16 maps still select the bf16 volume (BASELINE config 3)
    blk_b = hook(fmap1=f1.bfloat16().to(gpu), fmap2=f2.bfloat16().to(gpu), num_levels=4, radius=4)
    assert blk_b.volume_dtype == torch.bfloat16
