"""Backward of the correlation path and of the convex upsampling (`pfk_corr_lookup_bwd_f32`, the volume /
pyramid backward; BASELINE config 5): libpfk gradients against torch autograd in float64 through the CPU oracle, which is
the reference's arithmetic (raft/corr.py:13-64, raft/utils.py:67-75, raft/raft.py:112-123).

Tolerance: 2e-4 of each gradient's scale (fp32 sums over N = h*w pixels)."""
import pytest
import torch

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


def rel_close(a, b, tol=2e-4, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _coords(B, h, w, g, spread):
    c = O.coords_grid(B, h, w) + (torch.rand(B, 2, h, w, generator=g) * 2 - 1) * spread
    c[0, :, 0, 0] = torch.tensor([-30.0, 2.0])          # whole window outside on the left
    c[0, :, 1, 1] = torch.tensor([w + 1.5, h - 0.5])    # straddles the bottom-right corner
    c[0, :, 2, 2] = torch.tensor([3.0, 4.0])            # integer coordinates: zero fractional weights
    return c


@pytest.mark.parametrize("B,h,w,L,r", [(1, 16, 24, 4, 4), (2, 13, 17, 2, 4), (1, 23, 31, 4, 3), (1, 9, 12, 3, 1)])
def test_lookup_backward_kernel(gpu, B, h, w, L, r):
    """d(out) -> gradient of every pyramid level, including odd map sizes (row stride padded to 4), out-of-map windows
    (zero padding has zero gradient) and accumulation over two calls."""
    g = torch.Generator().manual_seed(3)
    N = h * w
    sizes, hl, wl = [], h, w
    for _ in range(L):
        sizes.append((hl, wl))
        hl, wl = hl // 2, wl // 2
    pyr = [torch.randn(B * N, 1, a, b, generator=g, dtype=torch.float64).requires_grad_() for a, b in sizes]
    n = 2 * r + 1
    total = 0
    cs, gos = [], []
    for rep in range(2):
        c = _coords(B, h, w, g, 5.0)
        out = O.lookup(pyr, c.double(), r)
        go = torch.randn(out.shape, generator=g, dtype=torch.float64)
        total = total + (out * go).sum()
        cs.append(c); gos.append(go)
    total.backward()
    lds = [max(4, (a * b + 3) // 4 * 4) for a, b in sizes]
    bufs = [torch.zeros(B * N, ld, device=gpu) for ld in lds]
    for c, go in zip(cs, gos):
        gpm = go.float().permute(0, 2, 3, 1).reshape(B * N, L * n * n).contiguous().to(gpu)
        torch.ops.pfk.corr_lookup_bwd(bufs, [s[0] for s in sizes], [s[1] for s in sizes], c.to(gpu), r, gpm)
    for buf, p, (a, b) in zip(bufs, pyr, sizes):
        rel_close(buf[:, : a * b].view(B * N, 1, a, b), p.grad, what=f"level {a}x{b}")
        assert bool((buf[:, a * b:] == 0).all())          # the pad columns stay zero (they are GEMM operands)


@pytest.mark.parametrize("mode,B,D,h,w,L,r", [("avgpool", 2, 64, 16, 24, 4, 4), ("avgpool", 1, 256, 23, 31, 4, 4),
                                              ("bilinear_f2", 1, 64, 16, 24, 4, 4), ("avgpool", 1, 32, 13, 11, 2, 3)])
def test_corr_block_autograd(gpu, mode, B, D, h, w, L, r):
    """`CorrBlock` under autograd: three lookups at different coordinates, gradients of fmap1 / fmap2 vs float64 autograd
    through the oracle's materialised pyramid (avg-pooled volume, or SEA-RAFT's per-level GEMM against bilinear-halved fmap2)."""
    from ptlflow_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn(B, D, h, w, generator=g)
    f2 = torch.randn(B, D, h, w, generator=g)
    cs = [_coords(B, h, w, g, 4.0) for _ in range(3)]
    n = 2 * r + 1
    gos = [torch.randn(B, L * n * n, h, w, generator=g) for _ in range(3)]
    # float64 oracle
    a, b = f1.double().requires_grad_(), f2.double().requires_grad_()
    pyr = (O.correlation_pyramid if mode == "avgpool" else O.sea_correlation_pyramid)(a, b, L)
    if mode != "avgpool":   # the oracle divides by a float32 sqrt there; redo the scale in float64
        pyr = [p * (torch.sqrt(torch.tensor(D).float()).double() / torch.sqrt(torch.tensor(float(D), dtype=torch.float64))) for p in pyr]
    tot = sum((O.lookup(pyr, c.double(), r) * go.double()).sum() for c, go in zip(cs, gos))
    tot.backward()
    # libpfk
    x1, x2 = f1.to(gpu).requires_grad_(), f2.to(gpu).requires_grad_()
    cb = CorrBlock(x1, x2, L, r, pyramid=mode)
    outs = [cb(c.to(gpu)) for c in cs]
    for o, c in zip(outs, cs):   # forward values unchanged by the autograd wrapping
        ref = O.lookup([p.detach().float() for p in pyr], c, r)
        assert (o.detach().cpu() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    sum((o * go.to(gpu)).sum() for o, go in zip(outs, gos)).backward()
    rel_close(x1.grad, a.grad, what="d fmap1")
    rel_close(x2.grad, b.grad, what="d fmap2")


@pytest.mark.parametrize("B,H,W", [(2, 9, 13), (1, 46, 62)])
def test_convex_upsample_backward(gpu, B, H, W):
    from ptlflow_amd.train import convex_upsample
    g = torch.Generator().manual_seed(11)
    flow = (torch.randn(B, 2, H, W, generator=g) * 3).double().requires_grad_()
    mask = torch.randn(B, 576, H, W, generator=g).double().requires_grad_()
    ref = O.convex_upsample(flow, mask)
    go = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(go)
    fg = flow.detach().float().to(gpu).requires_grad_()
    mg = mask.detach().float().permute(0, 2, 3, 1).reshape(B * H * W, 576).contiguous().to(gpu).requires_grad_()
    out = convex_upsample(fg, mg)
    rel_close(out, ref, tol=1e-5, what="forward")
    out.backward(go.float().to(gpu))
    rel_close(fg.grad, flow.grad, what="d flow")
    rel_close(mg.grad.view(B, H, W, 576).permute(0, 3, 1, 2), mask.grad, what="d mask")
