"""Seams B1 + B3 used the way the reference's loop uses them (raft.py:144-187): NCHW tensors in, NCHW out,
`get_corr_block(...)` once, then `corr_fn(coords1)` / `update_block(net, inp, corr, flow)` per iteration."""
import pytest
import torch

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("small", [False, True])
def test_reference_style_loop(gpu, small):
    from ptlflow_amd.corr import get_corr_block
    from ptlflow_amd.raft import _param_tree
    from ptlflow_amd.synth import synth_state_dict, update_block_shapes
    from ptlflow_amd.update import PfkUpdateBlock, basic_spec, small_spec

    spec = small_spec() if small else basic_spec()
    r = 3 if small else 4
    holder = _param_tree(update_block_shapes(spec))            # stands in for the reference's BasicUpdateBlock
    P = synth_state_dict({k: tuple(v.shape) for k, v in holder.state_dict().items()}, seed=9)
    holder.load_state_dict(P)
    ub = PfkUpdateBlock(holder, spec).cuda().eval()
    assert set(ub.state_dict()) == set(P)                       # checkpoint keys unchanged by the wrapper

    g = torch.Generator().manual_seed(3)
    B, D, h, w = 2, 64, 18, 26
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    net = torch.tanh(torch.randn(B, spec.hidden, h, w, generator=g))
    inp = torch.relu(torch.randn(B, spec.context, h, w, generator=g))

    # oracle loop
    pyr = O.correlation_pyramid(f1, f2, 4)
    c0 = O.coords_grid(B, h, w)
    c1 = c0.clone()
    n_ref = net
    step = O.small_update_block if small else O.basic_update_block
    for _ in range(4):
        corr = O.lookup(pyr, c1, r)
        n_ref, m_ref, d = step(P, n_ref, inp, corr, c1 - c0)
        c1 = c1 + d

    # reference-style loop on the GPU through the two seams
    with torch.no_grad():
        corr_fn = get_corr_block(fmap1=f1.cuda(), fmap2=f2.cuda(), radius=r, num_levels=4, alternate_corr=False)
        g0 = c0.cuda()
        g1 = g0.clone()
        n, i = net.cuda(), inp.cuda()
        for _ in range(4):
            g1 = g1.detach()
            corr = corr_fn(g1)
            assert tuple(corr.shape) == (B, spec.corr_channels, h, w)
            flow = g1 - g0
            n, up_mask, delta = ub(n, i, corr, flow)
            g1 = g1 + delta
    err = (g1.cpu() - c1).abs().max().item()
    assert err < 2e-4, f"coords differ by {err:.2e}"
    assert (n.cpu() - n_ref).abs().max().item() < 2e-4
    if small:
        assert up_mask is None
    else:
        assert tuple(up_mask.shape) == (B, 576, h, w)
        assert (up_mask.cpu() - m_ref).abs().max().item() < 2e-4


def test_sea_raft_pyramid_mode(gpu):
    """sea_raft/corr.py:71-84: per-level GEMM against bilinear-halved fmap2."""
    from ptlflow_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(4)
    f1, f2 = torch.randn(1, 64, 16, 24, generator=g), torch.randn(1, 64, 16, 24, generator=g)
    pyr = O.sea_correlation_pyramid(f1, f2, 4)
    cb = CorrBlock(f1.cuda(), f2.cuda(), 4, 4, pyramid="bilinear_f2")
    for a, b in zip(cb.corr_pyramid, pyr):
        assert (a.cpu().reshape(b.shape) - b).abs().max().item() < 2e-5
    c = O.coords_grid(1, 16, 24) + torch.rand(1, 2, 16, 24, generator=g) * 6 - 3
    ref = O.lookup(pyr, c, 4)
    assert (cb(c.cuda()).cpu() - ref).abs().max().item() < 5e-5


def test_sea_raft_pyramid_north_star_shape(gpu):
    """SEA-RAFT's pyramid at the north-star shape (55x128 grid, D = 256): per-level volume against `fmap2` halved by the
    HIP 2x2 feature-map average (== bilinear x0.5), lookups vs the oracle (sea_raft/corr.py:71-117)."""
    from ptlflow_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(14)
    f1, f2 = torch.randn(1, 256, 55, 128, generator=g), torch.randn(1, 256, 55, 128, generator=g)
    pyr = O.sea_correlation_pyramid(f1, f2, 4)
    cb = CorrBlock(f1.cuda(), f2.cuda(), 4, 4, pyramid="bilinear_f2")
    assert [tuple(p.shape[1:]) for p in cb.corr_pyramid] == [(55, 128), (27, 64), (13, 32), (6, 16)]
    for a, b in zip(cb.corr_pyramid, pyr):
        assert (a.cpu().reshape(b.shape) - b).abs().max().item() < 3e-5
    c = O.coords_grid(1, 55, 128) + torch.rand(1, 2, 55, 128, generator=g) * 10 - 5
    ref = O.lookup(pyr, c, 4)
    assert (cb(c.cuda()).cpu() - ref).abs().max().item() < 1e-4
    # the lookup of the kernel's OWN pyramid is bit-exact
    assert torch.equal(cb(c.cuda()).cpu(), O.lookup([p.cpu().unsqueeze(1) for p in cb.corr_pyramid], c, 4))


def test_sea_raft_model_golden(gpu):
    """Seam B1 as the real SEA-RAFT model uses it (tests/golden/sea_raft_model.pt: the feature maps of its `get_corr_block`
    call and every iteration's coords -> lookup, recorded from the live reference): `CorrBlock(pyramid="bilinear_f2")`."""
    import os
    from ptlflow_amd.corr import get_corr_block
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sea_raft_model.pt"))
    corr_fn = get_corr_block(gold["fmap1"].cuda(), gold["fmap2"].cuda(), num_levels=gold["levels"], radius=gold["radius"],
                             pyramid="bilinear_f2")
    for call in gold["calls"]:
        got = corr_fn(call["coords"].cuda()).cpu()
        assert got.shape == call["out"].shape
        assert (got - call["out"]).abs().max().item() < 2e-4 * max(1.0, call["out"].abs().max().item())


@pytest.mark.parametrize("B,H1,W1,H2,W2,C,r", [(1, 16, 24, 16, 24, 256, 4), (2, 11, 13, 5, 6, 128, 3), (1, 9, 10, 9, 10, 36, 4)])
def test_alt_cuda_corr_abi(gpu, B, H1, W1, H2, W2, C, r):
    """Seam B2: the module importable as `alt_cuda_corr` (correlation.cpp:23-37 contract) vs the oracle."""
    import importlib
    import ptlflow_amd.altcorr as altcorr
    altcorr.install()
    mod = importlib.import_module("alt_cuda_corr")
    g = torch.Generator().manual_seed(8)
    f1 = torch.randn(B, H1, W1, C, generator=g)
    f2 = torch.randn(B, H2, W2, C, generator=g)
    base = torch.stack(torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32),
                                      indexing="ij")[::-1], -1)[None, None].repeat(B, 1, 1, 1, 1)
    coords = base * (W2 / W1) + torch.randn(B, 1, H1, W1, 2, generator=g) * 3     # includes out-of-map windows
    coords[0, 0, 0, 0, 0] = float("nan")
    coords[0, 0, 0, 1, 1] = 1e12
    ref = O.alt_corr_forward(f1, f2, coords, r)
    (out,) = mod.forward(f1.cuda(), f2.cuda(), coords.cuda(), r)
    assert tuple(out.shape) == (B, 1, (2 * r + 1) ** 2, H1, W1)
    got = out.cpu()
    both_nan = torch.isnan(got) & torch.isnan(ref)
    err = torch.where(both_nan, torch.zeros_like(ref), (got - ref).abs())
    assert bool((torch.isnan(got) == torch.isnan(ref)).all())
    assert err.max().item() < 2e-4 * max(1.0, ref.nan_to_num().abs().max().item())
    with pytest.raises(RuntimeError):
        mod.forward(f1.cuda().permute(0, 2, 1, 3), f2.cuda(), coords.cuda(), r)   # CHECK_CONTIGUOUS


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("B,H1,W1,H2,W2,C,r,sigma", [
    (2, 27, 45, 27, 45, 256, 4, 0.6),     # smooth flow: every patch goes through the shared box (ragged patch grid)
    (1, 16, 24, 16, 24, 256, 4, 3.0),     # rough flow: most boxes overflow -> per-pixel fallback inside the kernel
    (2, 22, 13, 11, 6, 128, 3, 0.8),      # coarser fmap2 (pyramid level), r = 3, windows hanging over every border
    (1, 9, 10, 9, 10, 36, 4, 0.5),        # C = 36: two K-steps of 16 and one of 4
])
def test_altcorr_forward_kernels(gpu, mode, B, H1, W1, H2, W2, C, r, sigma):
    """K7 forward, each kernel forced (`debug_set_altcorr`): 1 = one wave per pixel, 2 / 3 = the window-sharing MFMA kernel on
    8x4 / 8x8 patches (shared bounding box, per-pixel fallback for overflowing boxes), against the oracle's restatement of
    correlation_kernel.cu:18-119; NaN / far-away coordinates keep their pattern."""
    g = torch.Generator().manual_seed(31 + mode)
    f1 = torch.randn(B, H1, W1, C, generator=g)
    f2 = torch.randn(B, H2, W2, C, generator=g)
    base = torch.stack(torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32),
                                      indexing="ij")[::-1], -1)[None, None].repeat(B, 1, 1, 1, 1)
    smooth = torch.nn.functional.interpolate(torch.randn(B, 2, 4, 5, generator=g) * 4, size=(H1, W1), mode="bicubic", align_corners=True)
    coords = (base + smooth.permute(0, 2, 3, 1)[:, None]) * (W2 / W1) + torch.randn(B, 1, H1, W1, 2, generator=g) * sigma
    coords[0, 0, 0, 0, 0] = float("nan")
    coords[0, 0, 0, 1, 1] = 1e12
    coords[0, 0, H1 - 1, W1 - 1] = torch.tensor([-30.0, 2.5])       # a window entirely left of the map
    ref = O.alt_corr_forward(f1, f2, coords, r)
    torch.ops.pfk.debug_set_altcorr(mode)
    try:
        out = torch.ops.pfk.altcorr_forward(f1.cuda(), f2.cuda(), coords.cuda(), r)
    finally:
        torch.ops.pfk.debug_set_altcorr(0)
    got = (out[0] if isinstance(out, (list, tuple)) else out).cpu()
    assert got.shape == ref.shape
    assert bool((torch.isnan(got) == torch.isnan(ref)).all())
    err = torch.where(torch.isnan(ref), torch.zeros_like(ref), (got - ref).abs())
    assert err.max().item() < 2e-4 * max(1.0, ref.nan_to_num().abs().max().item())


@pytest.mark.parametrize("B,C,H,W,L,r", [(1, 256, 16, 24, 4, 4), (2, 128, 24, 40, 2, 4), (1, 64, 18, 22, 3, 3)])
def test_alternate_corr_block(gpu, B, C, H, W, L, r):
    """`get_corr_block(alternate_corr=True)` (raft/corr.py:67-101, the default of ccmr / ms_raft_p): pooled fmap2 per level,
    on-demand windows, same channel layout and scaling as the oracle's restatement."""
    from ptlflow_amd.corr import AlternateCorrBlock, get_corr_block
    g = torch.Generator().manual_seed(12)
    f1, f2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    coords = O.coords_grid(B, H, W) + torch.randn(B, 2, H, W, generator=g) * 4
    ref = O.alternate_corr_block(f1, f2, coords, L, r)
    blk = get_corr_block(fmap1=f1.cuda(), fmap2=f2.cuda(), num_levels=L, radius=r, alternate_corr=True)
    assert isinstance(blk, AlternateCorrBlock)
    got = blk(coords.cuda()).cpu()
    assert got.shape == ref.shape == (B, L * (2 * r + 1) ** 2, H, W)
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_alternate_corr_block_vs_reference_golden(gpu, mode):
    """`get_corr_block(alternate_corr=True)` on libpfk against what the REFERENCE computes for it without its CUDA extension
    (`IterativeCorrBlock`, recorded by oracle/make_golden.py::golden_alt_corr; 64x136 grid: the library's own choice for level 0
    is the window-sharing kernel), with the library's choice and with each forward kernel forced."""
    import os
    from ptlflow_amd.corr import get_corr_block
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "alt_corr.pt"))
    torch.ops.pfk.debug_set_altcorr(mode)
    try:
        blk = get_corr_block(fmap1=gold["fmap1"].cuda(), fmap2=gold["fmap2"].cuda(), num_levels=gold["levels"], radius=gold["radius"],
                             alternate_corr=True)
        got = blk(gold["coords"].cuda()).cpu()[:, :, ::gold["row_step"]]
    finally:
        torch.ops.pfk.debug_set_altcorr(0)
    ref = gold["out_rows"]
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


def test_gma_update_block_dropin(gpu):
    """Seam B3 with GMA's five-argument forward(net, inp, corr, flow, attention) (gma/update.py:148)."""
    from ptlflow_amd.raft import _param_tree
    from ptlflow_amd.synth import synth_state_dict, update_block_shapes
    from ptlflow_amd.update import PfkUpdateBlock, gma_spec
    spec = gma_spec()
    holder = _param_tree(update_block_shapes(spec))
    P = synth_state_dict({k: tuple(v.shape) for k, v in holder.state_dict().items()}, seed=19)
    holder.load_state_dict(P)
    ub = PfkUpdateBlock(holder, spec).cuda().eval()
    g = torch.Generator().manual_seed(5)
    B, h, w = 2, 10, 14
    net = torch.tanh(torch.randn(B, 128, h, w, generator=g))
    inp = torch.relu(torch.randn(B, 128, h, w, generator=g))
    corr = torch.randn(B, 324, h, w, generator=g)
    flow = torch.randn(B, 2, h, w, generator=g) * 2
    attn = torch.softmax(torch.randn(B, 1, h * w, h * w, generator=g), dim=-1)
    n_ref, m_ref, d_ref = O.gma_update_block(P, net, inp, corr, flow, attn)
    with torch.no_grad():
        n, m, d = ub(net.cuda(), inp.cuda(), corr.cuda(), flow.cuda(), attn.cuda())
    for a, b in ((n, n_ref), (m, m_ref), (d, d_ref)):
        assert (a.cpu() - b).abs().max().item() < 2e-4


def test_alt_cuda_corr_backward(gpu):
    """`alt_cuda_corr.backward` (correlation.cpp:39-49) vs autograd through the oracle's restatement of the forward."""
    import ptlflow_amd.altcorr as altcorr
    g = torch.Generator().manual_seed(9)
    B, H1, W1, H2, W2, C, r = 2, 9, 11, 7, 8, 64, 3
    f1 = torch.randn(B, H1, W1, C, generator=g, requires_grad=True)
    f2 = torch.randn(B, H2, W2, C, generator=g, requires_grad=True)
    base = torch.stack(torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32),
                                      indexing="ij")[::-1], -1)[None, None].repeat(B, 1, 1, 1, 1)
    coords = base * 0.7 + torch.randn(B, 1, H1, W1, 2, generator=g) * 2
    out = O.alt_corr_forward(f1, f2, coords, r)
    grad = torch.randn(out.shape, generator=g)
    out.backward(grad)
    g1, g2, gc = altcorr.backward(f1.detach().cuda(), f2.detach().cuda(), coords.cuda(), grad.cuda(), r)
    assert (g1.cpu() - f1.grad).abs().max().item() < 1e-4 * max(1.0, f1.grad.abs().max().item())
    assert (g2.cpu() - f2.grad).abs().max().item() < 1e-4 * max(1.0, f2.grad.abs().max().item())
    assert tuple(gc.shape) == tuple(coords.shape) and float(gc.abs().max()) == 0.0


def test_ms_raft_plus_update_block(gpu):
    """MS-RAFT+'s update block (ms_raft_plus/update.py:119-153, stack_coords=False): RAFT's block on a 2-level lookup
    (162 correlation channels: not a multiple of 4) with a x2 mask head (36 channels), fed by `get_corr_block` as the model does."""
    from ptlflow_amd.corr import get_corr_block
    from ptlflow_amd.raft import _param_tree
    from ptlflow_amd.synth import synth_state_dict, update_block_shapes
    from ptlflow_amd.update import PfkUpdateBlock, ms_raft_plus_spec
    spec = ms_raft_plus_spec(162)
    holder = _param_tree(update_block_shapes(spec))
    P = synth_state_dict({k: tuple(v.shape) for k, v in holder.state_dict().items()}, seed=23)
    holder.load_state_dict(P)
    ub = PfkUpdateBlock(holder, spec).cuda().eval()
    g = torch.Generator().manual_seed(7)
    B, D, h, w = 2, 64, 14, 18
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    net = torch.tanh(torch.randn(B, 128, h, w, generator=g))
    inp = torch.relu(torch.randn(B, 128, h, w, generator=g))
    pyr = O.correlation_pyramid(f1, f2, 2)
    c0 = O.coords_grid(B, h, w)
    c1, n_ref = c0.clone(), net
    for _ in range(3):
        n_ref, m_ref, d = O.basic_update_block(P, n_ref, inp, O.lookup(pyr, c1, 4), c1 - c0)
        c1 = c1 + d
    with torch.no_grad():
        corr_fn = get_corr_block(fmap1=f1.cuda(), fmap2=f2.cuda(), radius=4, num_levels=2, alternate_corr=False)
        g0 = c0.cuda()
        g1, n, i = g0.clone(), net.cuda(), inp.cuda()
        for _ in range(3):
            corr = corr_fn(g1)
            assert tuple(corr.shape) == (B, 162, h, w)
            n, up_mask, delta = ub(n, i, corr, g1 - g0, None, None)        # coords_x / coords_y: unused when stack_coords=False
            g1 = g1 + delta
    assert (g1.cpu() - c1).abs().max().item() < 2e-4
    assert tuple(up_mask.shape) == (B, 36, h, w) and (up_mask.cpu() - m_ref).abs().max().item() < 2e-4
    assert (n.cpu() - n_ref).abs().max().item() < 2e-4


def test_ccmr_update_block_hybrid(gpu):
    """CCMR's update block (ccmr/update.py:110-168): motion encoder, SepConvGRU(512) and heads on the kernels around the
    block's OWN aggregator module (an XCiT per scale in the reference; any `aggregator[level](global_context, motion)` here),
    called at two scales one after the other as the coarse-to-fine loop does (ccmr.py:180-213)."""
    from ptlflow_amd.raft import _param_tree
    from ptlflow_amd.synth import synth_state_dict, update_block_shapes
    from ptlflow_amd.update import PfkUpdateBlock, ccmr_spec

    class Agg(torch.nn.Module):          # stands in for XCiT(embed_dim=128, separate=True): (context, motion) -> 128 channels
        def __init__(self):
            super().__init__()
            self.mix = torch.nn.Conv2d(256, 128, 1)

        def forward(self, ctx, mf):
            return torch.tanh(self.mix(torch.cat([ctx, mf], 1)))

    spec = ccmr_spec(162)
    holder = _param_tree(update_block_shapes(spec))
    P = synth_state_dict({k: tuple(v.shape) for k, v in holder.state_dict().items()}, seed=29)
    holder.load_state_dict(P)
    torch.manual_seed(3)
    holder.add_module("aggregator", torch.nn.ModuleList([Agg(), Agg()]))
    cpu_aggs = [a for a in holder.aggregator]
    import copy
    cpu_aggs = copy.deepcopy(cpu_aggs)
    ub = PfkUpdateBlock(holder, spec).cuda().eval()
    assert any(k.startswith("aggregator.") for k in ub.state_dict())       # the foreign sub-module stays registered
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for level, (h, w) in enumerate(((9, 12), (18, 24))):
            B = 1
            net = torch.tanh(torch.randn(B, 128, h, w, generator=g))
            inp = torch.relu(torch.randn(B, 128, h, w, generator=g))
            ctx = torch.randn(B, 128, h, w, generator=g)
            n_ref, n = net, net.cuda()
            for it in range(2):
                corr = torch.randn(B, 162, h, w, generator=g)
                flow = torch.randn(B, 2, h, w, generator=g) * 2
                n_ref, m_ref, d_ref = O.ccmr_update_block(P, n_ref, inp, corr, flow, cpu_aggs[level], ctx)
                n, m, d = ub(n, inp.cuda(), corr.cuda(), flow.cuda(), ctx.cuda(), level_index=level)
                for a, b in ((n, n_ref), (m, m_ref), (d, d_ref)):
                    assert (a.cpu() - b).abs().max().item() < 2e-4
            assert tuple(m.shape) == (B, 36, h, w)


def test_alt_cuda_corr_backward_multi_set_and_half_inputs(gpu):
    """`alt_cuda_corr.backward` with N = 2 coordinate sets per call (correlation_kernel.cu:122-256 loops over N) and `forward` on
    half-precision callers (the reference's callers up-cast, raft/corr.py:90-96; the module accepts them directly)."""
    import ptlflow_amd.altcorr as altcorr
    g = torch.Generator().manual_seed(10)
    B, H1, W1, H2, W2, C, r, N = 1, 8, 10, 8, 10, 64, 2, 2
    f1 = torch.randn(B, H1, W1, C, generator=g, requires_grad=True)
    f2 = torch.randn(B, H2, W2, C, generator=g, requires_grad=True)
    base = torch.stack(torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32),
                                      indexing="ij")[::-1], -1)[None, None].repeat(B, N, 1, 1, 1)
    coords = base + torch.randn(B, N, H1, W1, 2, generator=g) * 2
    outs = [O.alt_corr_forward(f1, f2, coords[:, n:n + 1], r) for n in range(N)]
    out = torch.cat(outs, 1)
    grad = torch.randn(out.shape, generator=g)
    out.backward(grad)
    g1, g2, gc = altcorr.backward(f1.detach().cuda(), f2.detach().cuda(), coords.cuda(), grad.cuda(), r)
    assert (g1.cpu() - f1.grad).abs().max().item() < 1e-4 * max(1.0, f1.grad.abs().max().item())
    assert (g2.cpu() - f2.grad).abs().max().item() < 1e-4 * max(1.0, f2.grad.abs().max().item())
    assert tuple(gc.shape) == tuple(coords.shape) and float(gc.abs().max()) == 0.0
    (fwd,) = altcorr.forward(f1.detach().cuda(), f2.detach().cuda(), coords.cuda(), r)
    assert (fwd.cpu() - out.detach()).abs().max().item() < 2e-4 * max(1.0, out.abs().max().item())
    (h,) = altcorr.forward(f1.detach().half().cuda(), f2.detach().half().cuda(), coords.half().cuda(), r)
    assert h.dtype == torch.float16
    ref16 = torch.cat([O.alt_corr_forward(f1.detach().half().float(), f2.detach().half().float(), coords.half().float()[:, n:n + 1], r)
                       for n in range(N)], 1)
    assert (h.float().cpu() - ref16).abs().max().item() < 2e-2 * max(1.0, ref16.abs().max().item())


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_altcorr_forward_bf16_kernels(gpu, mode):
    """`pfk_altcorr_forward_bf16` (SURVEY §8 f1 "fp32 + bf16"): bf16 feature maps widened exactly, fp32 products and
    accumulation — each forward kernel forced, against the oracle's fp32 restatement run on the SAME bf16-rounded values
    (so the only difference left is summation order), with NaN / far-away coordinates and a coarser fmap2."""
    for (B, H1, W1, H2, W2, C, r, sigma) in [(2, 27, 45, 27, 45, 256, 4, 0.6), (1, 16, 24, 16, 24, 256, 4, 3.0),
                                             (2, 22, 13, 11, 6, 128, 3, 0.8), (1, 9, 10, 9, 10, 36, 4, 0.5)]:
        g = torch.Generator().manual_seed(131 + mode)
        f1 = torch.randn(B, H1, W1, C, generator=g).bfloat16()
        f2 = torch.randn(B, H2, W2, C, generator=g).bfloat16()
        base = torch.stack(torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32),
                                          indexing="ij")[::-1], -1)[None, None].repeat(B, 1, 1, 1, 1)
        smooth = torch.nn.functional.interpolate(torch.randn(B, 2, 4, 5, generator=g) * 4, size=(H1, W1), mode="bicubic", align_corners=True)
        coords = (base + smooth.permute(0, 2, 3, 1)[:, None]) * (W2 / W1) + torch.randn(B, 1, H1, W1, 2, generator=g) * sigma
        coords[0, 0, 0, 0, 0] = float("nan")
        coords[0, 0, 0, 1, 1] = 1e12
        ref = O.alt_corr_forward(f1.float(), f2.float(), coords, r)
        torch.ops.pfk.debug_set_altcorr(mode)
        try:
            got = torch.ops.pfk.altcorr_forward(f1.cuda(), f2.cuda(), coords.cuda(), r).cpu()
        finally:
            torch.ops.pfk.debug_set_altcorr(0)
        assert got.dtype == torch.float32 and got.shape == ref.shape
        assert bool((torch.isnan(got) == torch.isnan(ref)).all())
        err = torch.where(torch.isnan(ref), torch.zeros_like(ref), (got - ref).abs())
        assert err.max().item() < 2e-4 * max(1.0, ref.nan_to_num().abs().max().item()), (mode, C, err.max().item())


def test_alternate_corr_block_bf16_vs_reference_autocast(gpu):
    """`AlternateCorrBlock` on bf16 feature maps (what autocast callers and `RAFT(conv_precision="bf16", alternate_corr=True)`
    hand it) against the REFERENCE's `IterativeCorrBlock` under `torch.autocast("cpu", bfloat16)` on the same maps
    (tests/golden/alt_corr_bf16.pt, oracle/make_golden.py::golden_alt_corr_bf16) and against the reference's fp32 output:
    the kernel multiplies the bf16 values exactly, so it must sit INSIDE the reference's own bf16 gap on both counts."""
    import os
    from ptlflow_amd.corr import AlternateCorrBlock, get_corr_block
    gd = os.path.join(os.path.dirname(__file__), "golden")
    gold, gbf = torch.load(os.path.join(gd, "alt_corr.pt")), torch.load(os.path.join(gd, "alt_corr_bf16.pt"))
    blk = get_corr_block(fmap1=gold["fmap1"].cuda().bfloat16(), fmap2=gold["fmap2"].cuda().bfloat16(), num_levels=gold["levels"],
                         radius=gold["radius"], alternate_corr=True)
    assert isinstance(blk, AlternateCorrBlock) and blk.map_dtype == torch.bfloat16
    out = blk(gold["coords"].cuda())
    assert out.dtype == torch.bfloat16                       # the caller's dtype, like the reference's half path (raft/corr.py:96)
    got = blk.lookup_pm(gold["coords"].cuda())               # fp32 values before that last rounding
    B, C, H, W = out.shape
    got = got.view(B, H, W, C).permute(0, 3, 1, 2).cpu()[:, :, ::gold["row_step"]]
    gap = gbf["autocast_gap_max"]
    e32 = (got - gold["out_rows"]).abs().max().item()
    eac = (got - gbf["out_rows_autocast"]).abs().max().item()
    print(f"bf16 on-demand correlation: max |err| vs the reference's fp32 output {e32:.3e}, vs its autocast output {eac:.3e} "
          f"(the reference's own autocast-vs-fp32 gap: {gap:.3e})")
    assert e32 <= gap and eac <= 2.0 * gap
