"""Parity of every libpfk kernel (through torch.ops.pfk -> C ABI) against the CPU oracle.

Tolerances (fp32):
  * lookup: BIT-EXACT for the same pyramid (tap indices and values) — north-star requirement;
  * pooling: bit-exact for the same input (same summation order);
  * GEMM-shaped kernels: |err| <= 2e-5 * (1 + |ref|) — fp32 MFMA accumulates a k-ordered fmaf chain,
    MKL/oneDNN use a different summation order; values are O(1..10).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu

EPI_LINEAR, EPI_GRU_ZR, EPI_GRU_Q = 0, 1, 2


def close(a, b, rtol=2e-5, atol=2e-5):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"


def pm(x):  # NCHW cpu -> [M, C] gpu
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda()


def unpm(x, B, H, W):
    return x.view(B, H, W, -1).permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("B,h,w,D", [(1, 16, 24, 64), (2, 9, 13, 128), (1, 55, 128, 256)])
def test_corr_volume(gpu, B, h, w, D):
    torch.manual_seed(0)
    f1, f2 = torch.randn(B, D, h, w), torch.randn(B, D, h, w)
    ref = O.all_pairs_correlation(f1, f2)
    N = h * w
    out = torch.empty(B, N, N, device=gpu)
    torch.ops.pfk.corr_volume(pm(f1).view(B, N, D), pm(f2).view(B, N, D), 1.0 / math.sqrt(D), out)
    close(out.view(B * N, 1, h, w), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("M,H,W", [(7, 55, 128), (64, 27, 64), (5, 13, 32), (3, 6, 16), (4, 3, 5), (2, 1, 2)])
def test_pool_bit_exact(gpu, M, H, W):
    torch.manual_seed(1)
    x = torch.randn(M, 1, H, W)
    ref = O.pool2x2(x)
    out = torch.empty(M, H // 2, W // 2, device=gpu)
    torch.ops.pfk.corr_pool2x2(x.view(M, H, W).cuda(), out)
    assert torch.equal(out.cpu(), ref.view(M, H // 2, W // 2))


def _coords_cases(B, h, w):
    g = torch.Generator().manual_seed(7)
    c0 = O.coords_grid(B, h, w)
    yield "integer", c0
    yield "fractional", c0 + torch.rand(B, 2, h, w, generator=g) * 16 - 8
    yield "half", c0 + 0.5
    yield "out_of_bounds", c0 + torch.randn(B, 2, h, w, generator=g) * 80
    edge = c0.clone()
    edge[:, 0] = torch.where(edge[:, 0] < w / 2, -0.25, w - 0.75)
    edge[:, 1] = torch.where(edge[:, 1] < h / 2, -1.0, float(h - 1))
    yield "edges", edge
    bad = c0.clone()
    bad[:, 0, 0, 0] = float("nan")
    bad[:, 1, 0, 1] = float("inf")
    bad[:, 0, 1, 0] = -float("inf")
    bad[:, 0, 1, 1] = 3.0e9
    bad[:, 1, 1, 2] = -2.5e7
    yield "nonfinite", bad


# (1, 47, 156, …) = KITTI 375×1242, the width with the most grid_sample round-trip index flips; (1, 46, 62, …) = Chairs
@pytest.mark.parametrize("B,h,w,L,r", [(1, 16, 24, 4, 4), (2, 23, 39, 4, 3), (1, 55, 128, 4, 4), (1, 8, 16, 4, 4), (1, 13, 17, 2, 4),
                                       (1, 47, 156, 4, 4), (1, 46, 62, 4, 4), (1, 47, 156, 4, 3)])
def test_lookup_bit_exact(gpu, B, h, w, L, r):
    torch.manual_seed(2)
    D = 32
    f1, f2 = torch.randn(B, D, h, w), torch.randn(B, D, h, w)
    pyr = O.correlation_pyramid(f1, f2, L)
    lv = [p.view(B * h * w, p.shape[-2], p.shape[-1]).contiguous().cuda() for p in pyr]
    n = 2 * r + 1
    for name, c in _coords_cases(B, h, w):
        ref = O.lookup(pyr, c, r)
        out = torch.full((B * h * w, L * n * n), -7.0, device=gpu)
        torch.ops.pfk.corr_lookup(lv, c.cuda(), r, out)
        got = unpm(out, B, h, w)
        same = (got == ref) | (torch.isnan(got) & torch.isnan(ref))
        assert bool(same.all()), f"{name}: {(~same).sum().item()} of {same.numel()} differ, max {(got - ref).abs().nan_to_num().max().item():.3e}"


def _to_blocked(p):
    """[M, (1,) h, w] map -> the blocked storage of include/pfk.h: ceil(h/4) x ceil(w/8) tiles of 4 x 8 elements, zero padded."""
    p = p.reshape(p.shape[0], p.shape[-2], p.shape[-1])
    M, h, w = p.shape
    th, tw = (h + 3) // 4, (w + 7) // 8
    q = torch.zeros(M, th * 4, tw * 8, dtype=p.dtype)
    q[:, :h, :w] = p
    return q.view(M, th, 4, tw, 8).permute(0, 1, 3, 2, 4).reshape(M, th * tw * 32).contiguous()


@pytest.mark.parametrize("B,h,w,L,r", [(1, 16, 24, 4, 4), (2, 23, 39, 4, 3), (1, 55, 128, 4, 4), (1, 8, 16, 4, 4), (1, 13, 17, 2, 4),
                                       (1, 47, 156, 4, 4), (1, 46, 62, 4, 4), (1, 47, 156, 4, 3)])
def test_lookup_blocked_bit_exact(gpu, B, h, w, L, r):
    """K3 on the blocked 4 x 8 volume layout (pfk_corr_lookup_blocked_f32): the oracle's bits on every coordinate family,
    and with 8 pixels per workgroup too."""
    torch.manual_seed(2)
    D = 32
    f1, f2 = torch.randn(B, D, h, w), torch.randn(B, D, h, w)
    pyr = O.correlation_pyramid(f1, f2, L)
    lv = [_to_blocked(p).cuda() for p in pyr]
    lh, lw = [p.shape[-2] for p in pyr], [p.shape[-1] for p in pyr]
    assert all(x.shape[1] == torch.ops.pfk.blocked_map_elems(a, b) for x, a, b in zip(lv, lh, lw))
    n = 2 * r + 1
    for pix in (104, 8):         # 104 = always 4 pixels per workgroup, 8 = always 8 (the default picks by the pixel count)
        torch.ops.pfk.debug_set_lookup_pix(pix)
        try:
            for name, c in _coords_cases(B, h, w):
                ref = O.lookup(pyr, c, r)
                out = torch.full((B * h * w, L * n * n), -7.0, device=gpu)
                torch.ops.pfk.corr_lookup_blocked(lv, lh, lw, c.cuda(), r, out)
                got = unpm(out, B, h, w)
                same = (got == ref) | (torch.isnan(got) & torch.isnan(ref))
                assert bool(same.all()), f"{name} (pix {pix}): {(~same).sum().item()} of {same.numel()} differ"
        finally:
            torch.ops.pfk.debug_set_lookup_pix(4)


@pytest.mark.parametrize("B,h,w,L,r", [(1, 16, 24, 4, 4), (2, 23, 39, 4, 3), (1, 55, 128, 4, 4), (1, 13, 17, 2, 4), (1, 47, 156, 4, 4), (1, 46, 62, 4, 2)])
def test_lookup_blocked_bf16_maps_bit_exact(gpu, B, h, w, L, r):
    """K3 on bf16 maps in the blocked layout (pfk_corr_lookup_blocked_bf16; BASELINE config 3's volume): the window is fetched as
    aligned pairs of bf16 elements — the result must be the oracle's lookup on the widened maps, bit for bit (grid_sample runs in fp32
    under autocast), on every coordinate family, odd level widths included (the pair's odd partner outside the map is zero padding)."""
    torch.manual_seed(3)
    D = 32
    f1, f2 = torch.randn(B, D, h, w), torch.randn(B, D, h, w)
    pyr = [p.to(torch.bfloat16) for p in O.correlation_pyramid(f1, f2, L)]
    wide = [p.float() for p in pyr]
    lh, lw = [p.shape[-2] for p in pyr], [p.shape[-1] for p in pyr]
    n = 2 * r + 1
    cases = [(name, c, O.lookup(wide, c, r)) for name, c in _coords_cases(B, h, w)]      # the oracle's answer, once per field
    for fill in (0.0, 9.0):       # whatever the pad elements of the edge tiles hold must not leak into a sample
        lv = []
        for p in pyr:
            q = p.reshape(p.shape[0], p.shape[-2], p.shape[-1])
            M_, hh, ww = q.shape
            th, tw = (hh + 3) // 4, (ww + 7) // 8
            t = torch.full((M_, th * 4, tw * 8), fill, dtype=torch.bfloat16)
            t[:, :hh, :ww] = q
            lv.append(t.view(M_, th, 4, tw, 8).permute(0, 1, 3, 2, 4).reshape(M_, th * tw * 32).contiguous().cuda())
        for pix in (104, 8):
            torch.ops.pfk.debug_set_lookup_pix(pix)
            try:
                for name, c, ref in cases:
                    out = torch.full((B * h * w, L * n * n), -7.0, device=gpu)
                    torch.ops.pfk.corr_lookup_blocked(lv, lh, lw, c.cuda(), r, out)
                    got = unpm(out, B, h, w)
                    same = (got == ref) | (torch.isnan(got) & torch.isnan(ref))
                    assert bool(same.all()), f"{name} (pix {pix}, pad {fill}): {(~same).sum().item()} of {same.numel()} differ"
            finally:
                torch.ops.pfk.debug_set_lookup_pix(4)


@pytest.mark.parametrize("M,H,W", [(7, 55, 128), (64, 27, 64), (5, 13, 32), (3, 6, 16), (4, 3, 5), (2, 1, 2), (3, 47, 156), (2, 23, 78)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_blocked_bit_exact(gpu, M, H, W, dtype):
    """K2 blocked -> blocked: torch's avg_pool2d bits, and zeros in the pad elements of the edge tiles."""
    torch.manual_seed(1)
    x = torch.randn(M, 1, H, W).to(dtype)
    if H // 2 == 0 or W // 2 == 0:
        ref = x.new_zeros(M, 1, H // 2, W // 2)          # a 1-pixel level pools to nothing (F.avg_pool2d refuses the shape)
    else:
        ref = F.avg_pool2d(x, 2, stride=2) if dtype == torch.bfloat16 else O.pool2x2(x)
    out = torch.full((M, torch.ops.pfk.blocked_map_elems(H // 2, W // 2)), 3.0, dtype=dtype, device=gpu)
    torch.ops.pfk.corr_pool2x2_blocked(_to_blocked(x).cuda(), out, H, W)
    if out.numel():
        assert torch.equal(out.cpu(), _to_blocked(ref))


def test_fmap_to_blocked(gpu):
    torch.manual_seed(4)
    for B, H, W, C in ((2, 55, 128, 256), (1, 47, 156, 128), (3, 5, 9, 32), (1, 1, 2, 64)):
        x = torch.randn(B * H * W, C)
        out = torch.full((B * torch.ops.pfk.blocked_map_elems(H, W), C), 9.0, device=gpu)
        torch.ops.pfk.fmap_to_blocked(x.cuda(), out, B, H, W)
        want = _to_blocked(x.view(B, H, W, C).permute(0, 3, 1, 2).reshape(B * C, H, W))          # [B*C, elems]
        want = want.view(B, C, -1).permute(0, 2, 1).reshape(-1, C)
        assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("B,h,w,D,L,r,dtype", [(1, 55, 128, 256, 4, 4, torch.float32), (2, 47, 156, 128, 4, 4, torch.float32),
                                                (1, 46, 62, 256, 2, 3, torch.float32), (1, 55, 128, 256, 4, 4, torch.bfloat16),
                                                (1, 9, 13, 64, 4, 4, torch.float32)])
@pytest.mark.parametrize("pyramid", ["avgpool", "bilinear_f2"])
def test_corr_block_layouts_agree(gpu, B, h, w, D, L, r, dtype, pyramid):
    """`CorrBlock(layout="blocked")` (the inference default) against `layout="rowmajor"`: the same pyramid (bit for bit: K1 against
    the permuted target map computes the same dot products) and the same lookups; row-major fp32 is the form the oracle tests pin."""
    from ptlflow_amd.corr import CorrBlock
    torch.manual_seed(6)
    f1, f2 = torch.randn(B, D, h, w, device=gpu).to(dtype), torch.randn(B, D, h, w, device=gpu).to(dtype)
    a = CorrBlock(f1, f2, num_levels=L, radius=r, pyramid=pyramid, layout="rowmajor")
    b = CorrBlock(f1, f2, num_levels=L, radius=r, pyramid=pyramid)
    assert a.layout == "rowmajor" and b.layout == "blocked" and b._levels[0].dim() == 2
    for pa, pb in zip(a.corr_pyramid, b.corr_pyramid):
        assert pa.shape == pb.shape and torch.equal(pa, pb)
    for name, c in _coords_cases(B, h, w):
        ra, rb = a(c.cuda()), b(c.cuda())
        same = (ra == rb) | (torch.isnan(ra) & torch.isnan(rb))
        assert bool(same.all()), name


@pytest.mark.parametrize("variant", [8, 14])
def test_lookup_variants_bit_exact(gpu, variant):
    """The two measurement variants of K3 (8 pixels per workgroup; cross-lane tap reads instead of the LDS patch,
    profiles/r04_e_lookup_shuffle.md) produce the oracle's bits too — the timing table compares equal results."""
    torch.manual_seed(3)
    B, h, w, L, r, D = 1, 23, 39, 4, 4, 32
    f1, f2 = torch.randn(B, D, h, w), torch.randn(B, D, h, w)
    pyr = O.correlation_pyramid(f1, f2, L)
    lv = [p.view(B * h * w, p.shape[-2], p.shape[-1]).contiguous().cuda() for p in pyr]
    n = 2 * r + 1
    torch.ops.pfk.debug_set_lookup_pix(variant)
    try:
        for name, c in _coords_cases(B, h, w):
            ref = O.lookup(pyr, c, r)
            out = torch.full((B * h * w, L * n * n), -7.0, device=gpu)
            torch.ops.pfk.corr_lookup(lv, c.cuda(), r, out)
            got = unpm(out, B, h, w)
            same = (got == ref) | (torch.isnan(got) & torch.isnan(ref))
            assert bool(same.all()), f"{name}: {(~same).sum().item()} of {same.numel()} differ"
    finally:
        torch.ops.pfk.debug_set_lookup_pix(4)


def _packed(weight, segs):
    from ptlflow_amd.packing import pack_conv_weight
    return pack_conv_weight(weight, segs).cuda()


@pytest.mark.parametrize("B,H,W,cin,cout,kh,kw,relu", [
    (1, 12, 20, 64, 96, 3, 3, True),
    (2, 9, 7, 324, 256, 1, 1, True),
    (1, 17, 33, 128, 126, 3, 3, True),
    (1, 10, 12, 256, 576, 1, 1, False),
    (1, 11, 19, 36, 40, 5, 1, False),
    (1, 55, 128, 128, 64, 3, 3, True),
])
def test_conv_linear(gpu, B, H, W, cin, cout, kh, kw, relu):
    torch.manual_seed(3)
    x = torch.randn(B, cin, H, W)
    wt = torch.randn(cout, cin, kh, kw) / math.sqrt(cin * kh * kw)
    bias = torch.randn(cout)
    ref = F.conv2d(x, wt, bias, padding=(kh // 2, kw // 2))
    if relu:
        ref = F.relu(ref)
    ref = ref * 0.25
    M = B * H * W
    buf = torch.full((M, cout + 12), 5.0, device=gpu)
    out = buf[:, 4:4 + cout]
    torch.ops.pfk.conv2d([pm(x)], B, H, W, kh, kw, _packed(wt, [(0, cin, cin)]), bias.cuda(), cout, EPI_LINEAR, relu, 0.25,
                         out, None, None, None)
    close(unpm(out, B, H, W), ref)
    assert bool((buf[:, :4] == 5.0).all()) and bool((buf[:, 4 + cout:] == 5.0).all()), "wrote outside its channel slice"


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 10])
def test_conv_all_tile_configs(gpu, tile):
    torch.manual_seed(4)
    B, H, W, cin, cout = 1, 21, 37, 96, 160
    x = torch.randn(B, cin, H, W)
    wt = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
    bias = torch.randn(cout)
    ref = F.conv2d(x, wt, bias, padding=1)
    out = torch.zeros(B * H * W, cout, device=gpu)
    torch.ops.pfk.debug_set_tile(tile)
    try:
        torch.ops.pfk.conv2d([pm(x)], B, H, W, 3, 3, _packed(wt, [(0, cin, cin)]), bias.cuda(), cout, EPI_LINEAR, False, 1.0,
                             out, None, None, None)
    finally:
        torch.ops.pfk.debug_set_tile(-1)
    close(unpm(out, B, H, W), ref)


@pytest.mark.parametrize("B,H,W,segs,cout,kh,kw", [
    (1, 12, 20, [64], 96, 3, 3),          # 8 tiles x 18 steps over 24 blocks: every tile split across 3 blocks
    (1, 55, 128, [128, 256], 128, 1, 5),  # the q conv of the headline config (220 tiles, 60 steps, 512 blocks)
    (2, 9, 7, [324], 256, 1, 1),          # short K (11 steps), partial channel chunk
    (1, 23, 31, [96, 148, 32], 160, 3, 3),
    (1, 5, 6, [32], 40, 1, 1),            # one K-step, one tile
])
def test_conv_stream_k(gpu, B, H, W, segs, cout, kh, kw):
    """Stream-K schedule (workspace given) vs the tile-grid schedule vs the CPU reference; and run-to-run determinism."""
    torch.manual_seed(13)
    cin = sum(segs)
    xs = [torch.randn(B, c, H, W) for c in segs]
    wt = torch.randn(cout, cin, kh, kw) / math.sqrt(cin * kh * kw)
    bias = torch.randn(cout)
    ref = F.conv2d(torch.cat(xs, 1), wt, bias, padding=(kh // 2, kw // 2))
    offs, o = [], 0
    for c in segs:
        offs.append((o, c, c)); o += c
    packed = _packed(wt, offs)
    srcs = [pm(x) for x in xs]
    M = B * H * W
    ws = torch.zeros(torch.ops.pfk.conv_workspace_bytes(), device=gpu, dtype=torch.uint8)
    outs = []
    for w in (ws, ws, None):
        out = torch.zeros(M, cout, device=gpu)
        torch.ops.pfk.conv2d(srcs, B, H, W, kh, kw, packed, bias.cuda(), cout, EPI_LINEAR, False, 1.0, out, None, None, None, w)
        outs.append(out)
    close(unpm(outs[0], B, H, W), ref)
    close(unpm(outs[2], B, H, W), ref)
    assert torch.equal(outs[0], outs[1]), "stream-K result is not deterministic"


def test_conv_multi_source(gpu):
    """torch.cat([a, b, c]) -> conv  ==  three channel-slice sources (one of them a strided slice)."""
    torch.manual_seed(5)
    B, H, W = 1, 14, 18
    ca, cb, cc, cout = 96, 148, 32, 128
    a, b, c = torch.randn(B, ca, H, W), torch.randn(B, cb, H, W), torch.randn(B, cc, H, W)
    b[:, 146:] = 0  # two zero pad channels, as in raft_small's hx
    wt = torch.randn(cout, ca + 146 + cc, 3, 3) / 40
    bias = torch.randn(cout)
    ref = F.conv2d(torch.cat([a, b[:, :146], c], 1), wt, bias, padding=1)
    wide = torch.zeros(B * H * W, 300, device=gpu)
    wide[:, 100:248] = pm(b)
    packed = _packed(wt, [(0, ca, ca), (ca, 146, 148), (ca + 146, cc, cc)])
    out = torch.zeros(B * H * W, cout, device=gpu)
    torch.ops.pfk.conv2d([pm(a), wide[:, 100:248], pm(c)], B, H, W, 3, 3, packed, bias.cuda(), cout, EPI_LINEAR, False, 1.0,
                         out, None, None, None)
    close(unpm(out, B, H, W), ref)


def _gru_params(Ch, Cx, passes):
    P = {}
    for kh, kw, sfx in passes:
        for k in "zrq":
            P[f"gru.conv{k}{sfx}.weight"] = torch.randn(Ch, Ch + Cx, kh, kw) / math.sqrt((Ch + Cx) * kh * kw)
            P[f"gru.conv{k}{sfx}.bias"] = torch.randn(Ch) * 0.1
    return P


@pytest.mark.parametrize("B,H,W,Ch,Cx,passes", [
    (1, 12, 16, 128, 256, ((1, 5, "1"), (5, 1, "2"))),   # SepConvGRU (raft)
    (1, 9, 21, 128, 384, ((1, 5, "1"), (5, 1, "2"))),    # SepConvGRU (gma/ccmr C_in=512)
    (2, 10, 14, 96, 148, ((3, 3, ""),)),                 # ConvGRU (raft_small, x padded 146->148)
])
def test_gru(gpu, B, H, W, Ch, Cx, passes):
    torch.manual_seed(6)
    P = _gru_params(Ch, Cx, passes)
    h = torch.tanh(torch.randn(B, Ch, H, W))
    x = torch.randn(B, Cx, H, W)
    ref = O.sepconv_gru(P, h, x) if len(passes) == 2 else O.conv_gru(P, h, x)
    from ptlflow_amd.packing import pack_conv_weight
    M = B * H * W
    hx = torch.cat([pm(h), pm(x)], 1).contiguous()
    z = torch.zeros(M, Ch, device=gpu)
    rh = torch.zeros(M, Ch, device=gpu)
    for kh, kw, sfx in passes:
        wzr = pack_conv_weight(torch.cat([P[f"gru.convz{sfx}.weight"], P[f"gru.convr{sfx}.weight"]], 0), [(0, Ch + Cx, Ch + Cx)]).cuda()
        bzr = torch.cat([P[f"gru.convz{sfx}.bias"], P[f"gru.convr{sfx}.bias"]]).cuda()
        wq = pack_conv_weight(P[f"gru.convq{sfx}.weight"], [(0, Ch, Ch), (Ch, Cx, Cx)]).cuda()
        bq = P[f"gru.convq{sfx}.bias"].cuda()
        torch.ops.pfk.conv2d([hx], B, H, W, kh, kw, wzr, bzr, 2 * Ch, EPI_GRU_ZR, False, 1.0, None, hx[:, :Ch], z, rh)
        torch.ops.pfk.conv2d([rh, hx[:, Ch:]], B, H, W, kh, kw, wq, bq, Ch, EPI_GRU_Q, False, 1.0, None, hx[:, :Ch], z, None)
    close(unpm(hx[:, :Ch], B, H, W), ref, rtol=2e-5, atol=3e-5)


@pytest.mark.parametrize("k,cout,W", [(7, 128, 17), (7, 64, 128), (3, 160, 33), (9, 32, 20)])
def test_conv_cin2(gpu, k, cout, W):
    torch.manual_seed(8)
    B, H = 2, 13
    flow = torch.randn(B, 2, H, W) * 3
    wt = torch.randn(cout, 2, k, k) / 10
    bias = torch.randn(cout)
    ref = F.relu(F.conv2d(flow, wt, bias, padding=k // 2))
    from ptlflow_amd.packing import pack_cin2_weight
    buf = torch.zeros(B * H * W, 384, device=gpu)
    buf[:, 382:384] = pm(flow)
    out = torch.zeros(B * H * W, cout, device=gpu)
    torch.ops.pfk.conv_cin2(buf[:, 382:384], pack_cin2_weight(wt).cuda(), bias.cuda(), out, B, H, W, k, True)
    close(unpm(out, B, H, W), ref)


@pytest.mark.parametrize("B,H,W,cout", [(2, 13, 17, 128), (1, 55, 128, 128), (1, 47, 156, 128), (2, 9, 70, 64), (1, 3, 5, 128)])
def test_conv_cin2_mfma_has_the_valu_kernels_bits(gpu, B, H, W, cout):
    """convf1 (7x7, 2 -> 64 / 128, raft/update.py:80,98) on the matrix cores: the accumulators start at the bias and the MFMA adds the
    98 products in the tiled VALU kernel's order (tap-major, x then y) — an fmaf chain: the two kernels must agree bit for bit, with
    fp32 and with bf16 rows out, into a channel slice of a wider buffer; and both match the oracle's convolution."""
    from ptlflow_amd.packing import pack_cin2_weight
    torch.manual_seed(8)
    flow = torch.randn(B, 2, H, W) * 3
    wt = torch.randn(cout, 2, 7, 7) / 10
    bias = torch.randn(cout)
    ref = F.relu(F.conv2d(flow, wt, bias, padding=3))
    buf = torch.zeros(B * H * W, 8, device=gpu)
    buf[:, 4:6] = pm(flow)
    w = pack_cin2_weight(wt).cuda()
    for dt in (torch.float32, torch.bfloat16):
        outs = []
        for valu in (1, 2):        # 1 = the tiled VALU kernel, 2 = the MFMA kernel whatever the size
            torch.ops.pfk.debug_set_cin2_valu(valu)
            try:
                wide = torch.full((B * H * W, cout + 8), -2.0, device=gpu, dtype=dt)
                torch.ops.pfk.conv_cin2(buf[:, 4:6], w, bias.cuda(), wide[:, 4: 4 + cout], B, H, W, 7, True)
            finally:
                torch.ops.pfk.debug_set_cin2_valu(0)
            assert bool((wide[:, :4] == -2.0).all()) and bool((wide[:, 4 + cout:] == -2.0).all()), "wrote outside the channel slice"
            outs.append(wide[:, 4: 4 + cout].float())
        assert torch.equal(outs[0], outs[1]), f"{dt}: max diff {(outs[0] - outs[1]).abs().max().item():.3e}"
        if dt == torch.float32:
            close(unpm(outs[1].contiguous(), B, H, W), ref)


@pytest.mark.parametrize("B,H,W,cin", [
    (2, 11, 15, 256),       # 4 pixels per wave, ragged row ends (15 = 3 waves + 3 pixels)
    (2, 48, 350, 256),      # >= 4096 waves of 8 pixels: the 8-pixel kernel, last wave of a row holds 6 pixels
    (1, 5, 9, 128),         # raft_small's flow head width
])
def test_flow_delta(gpu, B, H, W, cin):
    torch.manual_seed(9)
    x = torch.randn(B, cin, H, W)
    wt = torch.randn(2, cin, 3, 3) / 48
    bias = torch.randn(2)
    delta_ref = F.conv2d(x, wt, bias, padding=1)
    c0 = O.coords_grid(B, H, W)
    c1 = c0 + torch.randn(B, 2, H, W)
    from ptlflow_amd.packing import pack_flow_head_weight
    fm = torch.zeros(B * H * W, 512, device=gpu)
    fm[:, :cin] = pm(x)
    c1g = c1.clone().cuda()
    delta = torch.zeros(B, 2, H, W, device=gpu)
    hx = torch.zeros(B * H * W, 384, device=gpu)
    torch.ops.pfk.flow_delta(fm[:, :cin], pack_flow_head_weight(wt).cuda(), bias.cuda(), c0.cuda(), c1g, delta, hx[:, 382:384])
    close(delta, delta_ref)
    d = delta.cpu()
    assert torch.equal(c1g.cpu(), c1 + d)                       # coords1 += delta, one rounding
    assert torch.equal(unpm(hx[:, 382:384], B, H, W), (c1 + d) - c0)   # flow = coords1 - coords0
    f2 = torch.zeros(B * H * W, 4, device=gpu)
    torch.ops.pfk.flow_from_coords(c0.cuda(), c1.cuda(), f2[:, 1:3])
    assert torch.equal(unpm(f2[:, 1:3], B, H, W), c1 - c0)


def test_convex_upsample(gpu):
    torch.manual_seed(10)
    B, H, W = 2, 9, 13
    flow = torch.randn(B, 2, H, W) * 4
    mask = torch.randn(B, 576, H, W)
    ref = O.convex_upsample(flow, mask)
    out = torch.zeros(B, 2, 8 * H, 8 * W, device=gpu)
    torch.ops.pfk.convex_upsample(flow.cuda(), pm(mask), out)
    close(out, ref, rtol=1e-5, atol=1e-5)
    out2 = torch.zeros_like(out)
    hx = torch.zeros(B * H * W, 384, device=gpu)
    hx[:, 382:384] = pm(flow)
    torch.ops.pfk.convex_upsample_pm(hx[:, 382:384], pm(mask), out2)
    assert torch.equal(out, out2)
    # rows that are not 16-byte aligned take the one-pixel-per-wave kernel: same operation sequence, same bits as the
    # four-pixels-per-wave kernel above
    wide = torch.zeros(B * H * W, 577, device=gpu)
    wide[:, :576] = pm(mask)
    out3 = torch.zeros_like(out)
    torch.ops.pfk.convex_upsample(flow.cuda(), wide[:, :576], out3)
    assert torch.equal(out, out3)


@pytest.mark.parametrize("B,H,W", [(2, 9, 13), (1, 55, 128), (1, 1, 5)])
def test_upflow8(gpu, B, H, W):
    """raft_small's upsampling (raft/utils.py:94-96): 8 * F.interpolate(flow, 8x, bilinear, align_corners=True) on CPU."""
    torch.manual_seed(12)
    c0 = torch.randn(B, 2, H, W) * 30
    c1 = c0 + torch.randn(B, 2, H, W) * 4
    ref = 8 * F.interpolate(c1 - c0, size=(8 * H, 8 * W), mode="bilinear", align_corners=True)
    out = torch.zeros(B, 2, 8 * H, 8 * W, device=gpu)
    torch.ops.pfk.upflow8(c0.cuda(), c1.cuda(), out)
    close(out, ref, rtol=1e-5, atol=2e-5)


def test_layout_roundtrip(gpu):
    torch.manual_seed(11)
    x = torch.randn(2, 126, 7, 45)
    buf = torch.zeros(2 * 7 * 45, 384, device=gpu)
    torch.ops.pfk.nchw_to_pm(x.cuda(), buf[:, 256:382])
    assert torch.equal(buf[:, 256:382].cpu(), pm(x).cpu())
    back = torch.zeros(2, 126, 7, 45, device=gpu)
    torch.ops.pfk.pm_to_nchw(buf[:, 256:382], back)
    assert torch.equal(back.cpu(), x)


@pytest.mark.parametrize("small", [False, True])
def test_update_engine_step(gpu, small):
    """One whole BasicUpdateBlock / SmallUpdateBlock step vs the oracle (update.py:144-153 / :122-128)."""
    from ptlflow_amd.synth import synth_update_block_params
    from ptlflow_amd.update import UpdateEngine, basic_spec, small_spec
    torch.manual_seed(12)
    spec = small_spec() if small else basic_spec()
    P = synth_update_block_params(spec, seed=5)
    B, H, W = 1, 14, 22
    net = torch.tanh(torch.randn(B, spec.hidden, H, W))
    inp = torch.relu(torch.randn(B, spec.context, H, W))
    corr = torch.randn(B, spec.corr_channels, H, W)
    c0 = O.coords_grid(B, H, W)
    c1 = c0 + torch.randn(B, 2, H, W) * 2
    flow = c1 - c0
    step = O.small_update_block if small else O.basic_update_block
    net_ref, mask_ref, delta_ref = step(P, net, inp, corr, flow)
    eng = UpdateEngine(P, spec, gpu)
    eng.bind(B, H, W)
    eng.load_state(net.cuda(), inp.cuda())
    c0g, c1g = c0.cuda(), c1.clone().cuda()
    torch.ops.pfk.flow_from_coords(c0g, c1g, eng.flow_view)
    eng.step(pm(corr), c0g, c1g)
    close(eng.net_nchw(), net_ref, rtol=5e-5, atol=5e-5)
    close(c1g, c1 + delta_ref, rtol=5e-5, atol=5e-5)
    if not small:
        close(eng.mask_nchw(), mask_ref, rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("B,H,W", [(1, 55, 128), (2, 13, 17), (1, 8, 9), (3, 47, 156)])
def test_mask_upsample_fused(gpu, B, H, W):
    """pfk_mask_upsample_f32 (mask conv2 + softmax + convex upsampling, no mask in memory) against the two kernels it replaces
    (`conv2d` with scale 0.25 -> [M, 576] mask -> `convex_upsample_pm`): bit-identical; and against the oracle's
    `convex_upsample` on the oracle's mask (raft/raft.py:112-123, raft/update.py:152) to the GEMM tolerance."""
    from ptlflow_amd.packing import pack_conv_weight, permute_mask_head
    torch.manual_seed(9)
    M, cin = B * H * W, 256
    fm = torch.randn(M, 512)                       # fh | mask hidden: the kernel reads the second half as a strided view
    wt = torch.randn(576, cin, 1, 1) / math.sqrt(cin)
    bias = torch.randn(576) * 0.1
    hx = torch.randn(M, 8)
    flow_pm = hx[:, 4:6]                           # a 2-channel slice of a wider pixel-major row, like the engine's hx
    packed = pack_conv_weight(wt, [(0, cin, cin)])
    wp, bp = permute_mask_head(packed, bias)
    fm_g, hx_g = fm.cuda(), hx.cuda()
    x_g, flow_g = fm_g[:, 256:], hx_g[:, 4:6]
    mask = torch.empty(M, 576, device=gpu)
    torch.ops.pfk.conv2d([x_g], B, H, W, 1, 1, packed.cuda(), bias.cuda(), 576, EPI_LINEAR, False, 0.25, mask, None, None, None, None)
    want = torch.empty(B, 2, 8 * H, 8 * W, device=gpu)
    torch.ops.pfk.convex_upsample_pm(flow_g, mask, want)
    got = torch.full((B, 2, 8 * H, 8 * W), 7.0, device=gpu)
    torch.ops.pfk.mask_upsample(x_g, wp.cuda(), bp.cuda(), 0.25, flow_g, got)
    assert torch.equal(got, want), f"max diff {(got - want).abs().max().item():.3e}"
    x_nchw = fm[:, 256:].reshape(B, H, W, cin).permute(0, 3, 1, 2)
    mask_ref = 0.25 * F.conv2d(x_nchw, wt, bias)
    flow_ref = flow_pm.reshape(B, H, W, 2).permute(0, 3, 1, 2)
    close(got, O.convex_upsample(flow_ref, mask_ref), rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("B,H,W,with_ws", [(1, 55, 128, True), (1, 55, 128, False), (8, 55, 128, True), (2, 23, 31, True), (1, 46, 62, True)])
def test_conv_cout_active_has_the_full_launch_bits(gpu, B, H, W, with_ws):
    """`pfk_conv_desc.cout_active`: the fused flow-head | mask-head convolution (3x3, 128 -> 512) computing only its first 256 output
    channels.  Whatever schedule the full launch gets — at 1 x 55 x 128 with a workspace the stream-K split of the 880-tile grid,
    otherwise tile grids of 64x64 / 64x128 tiles — the active half must carry EXACTLY the full launch's bits and the other half of
    `out` must be left alone."""
    torch.manual_seed(11)
    cin, cout, act = 128, 512, 256
    M = B * H * W
    x = torch.randn(M, cin, device=gpu)
    wt = torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin)
    bias = (torch.randn(cout) * 0.1).cuda()
    packed = _packed(wt, [(0, cin, cin)])
    ws = torch.zeros(torch.ops.pfk.conv_workspace_bytes(), device=gpu, dtype=torch.uint8) if with_ws else None
    full = torch.empty(M, cout, device=gpu)
    for split in (0, act):      # `cout_split`: the two halves' column tiles interleaved in the stream-K order, in both launches
        torch.ops.pfk.conv2d([x], B, H, W, 3, 3, packed, bias, cout, EPI_LINEAR, True, 1.0, full, None, None, None, ws, None, 1, False, 0, split)
        half = torch.full((M, cout), -3.0, device=gpu)
        torch.ops.pfk.conv2d([x], B, H, W, 3, 3, packed, bias, cout, EPI_LINEAR, True, 1.0, half, None, None, None, ws, None, 1, False, act, split)
        assert torch.equal(half[:, :act], full[:, :act]), f"cout_split={split}"
    assert bool((half[:, act:] == -3.0).all())
    if ws is not None:      # the flag region is all zero again: skipped tiles neither publish nor consume
        off = torch.ops.pfk.conv_workspace_fault_offset()
        assert int(ws[off:off + 4].view(torch.int32).item()) == 0
        assert bool((ws[torch.ops.pfk.conv_workspace_bytes() - 768 * 64:][:768 * 4] == 0).all())
    ref = F.relu(F.conv2d(x.view(B, H, W, cin).permute(0, 3, 1, 2).cpu(), wt, bias.cpu(), padding=1))
    close(unpm(half[:, :act].contiguous(), B, H, W), ref[:, :act])


@pytest.mark.parametrize("B,H,W", [(1, 55, 128), (2, 13, 17), (1, 8, 9)])
@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_conv2d_group_equals_single_launches(gpu, B, H, W, n):
    """pfk_conv2d_group_f32: n independent LINEAR convolutions in one grid of 64 x 64 tiles — convc1 (1x1, 324 -> 256), convf2 (3x3,
    128 -> 64), the mask head's conv2 (1x1, 256 -> 576, x 0.25, no relu), a 3x3 192 -> 96 — against the same convolutions launched
    one by one without a workspace (bit for bit) and against the oracle's fp32 convolution (2e-5)."""
    from ptlflow_amd.packing import pack_conv_weight
    torch.manual_seed(11)
    M = B * H * W
    shapes = [(324, 256, 1, 1, 1.0), (128, 64, 3, 1, 1.0), (256, 576, 1, 0, 0.25), (192, 96, 3, 1, 1.0)][:n]
    srcs, ws, bs, outs, refs, ks, relus, scales = [], [], [], [], [], [], [], []
    for cin, cout, k, relu, scale in shapes:
        x = torch.randn(B, cin, H, W)
        wt = torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)
        bias = torch.randn(cout) * 0.1
        ref = F.conv2d(x, wt, bias, padding=k // 2)
        refs.append((F.relu(ref) if relu else ref) * scale)
        buf = torch.zeros(M, cin + 4, device=gpu)            # a view with a wider row stride, as the engine's slices are
        buf[:, :cin] = pm(x)
        srcs.append(buf[:, :cin]); ws.append(pack_conv_weight(wt, [(0, cin, cin)]).cuda()); bs.append(bias.cuda())
        outs.append(torch.full((M, cout + 8), -3.0, device=gpu)[:, :cout]); ks.append(k); relus.append(relu); scales.append(scale)
    torch.ops.pfk.conv2d_group(srcs, B, H, W, ks, ws, bs, relus, scales, outs, [])
    for i, (cin, cout, k, relu, scale) in enumerate(shapes):
        one = torch.full((M, cout), -5.0, device=gpu)
        torch.ops.pfk.conv2d([srcs[i]], B, H, W, k, k, ws[i], bs[i], cout, EPI_LINEAR, bool(relu), scale, one, None, None, None, None, None)
        assert torch.equal(outs[i], one), f"problem {i} differs from its single launch"
        close(unpm(outs[i], B, H, W), refs[i])
        assert bool((outs[i]._base[:, cout:] == -3.0).all()), "wrote past the output view"


def test_conv2d_group_with_residuals(gpu):
    """The grouped launch's `residuals` (out = residual + scale * (conv + bias)): GMA's per-pair `mf + gamma * attn @ v`
    (gma/gma_utils.py:100-113) — four independent [N x N] x [N x 128] products in one grid, each with its own A, weight, residual and
    output rows — against the same products launched one by one (bit for bit) and the oracle's matmul."""
    torch.manual_seed(12)
    H, W, C, n = 12, 20, 128, 4
    N = H * W
    attn = torch.softmax(torch.randn(n, N, N), -1)
    v = torch.randn(n, N, C)
    mf = torch.randn(n * N, 2 * C + 8)
    gamma = 0.37
    Np = (N + 31) // 32 * 32
    vT = torch.zeros(n, C, Np)
    vT[:, :, :N] = v.transpose(1, 2)
    hx = mf.cuda()
    hx1 = hx.clone()
    a_g, vT_g = attn.cuda(), vT.cuda()
    empty = torch.empty(0, device=gpu)
    rows = [slice(b * N, (b + 1) * N) for b in range(n)]
    torch.ops.pfk.conv2d_group([a_g[b] for b in range(n)], 1, H, W, [1] * n, [vT_g[b] for b in range(n)], [empty] * n, [0] * n, [gamma] * n,
                               [hx[r, C: 2 * C] for r in rows], [hx[r, :C] for r in rows])
    for b, r in enumerate(rows):
        torch.ops.pfk.conv2d([a_g[b]], 1, H, W, 1, 1, vT_g[b], None, C, EPI_LINEAR, False, gamma, hx1[r, C: 2 * C], None, None, None, None, hx1[r, :C])
    assert torch.equal(hx, hx1)
    ref = mf[:, :C].view(n, N, C) + gamma * torch.bmm(attn, v)
    close(hx[:, C: 2 * C].cpu().view(n, N, C), ref)
    assert torch.equal(hx[:, 2 * C:].cpu(), mf[:, 2 * C:]) and torch.equal(hx[:, :C].cpu(), mf[:, :C])
