"""The persistent cross-tile-pipelined stream-K convolution kernel (`conv_gemm_pp_kernel`, pfk_gemm.hip variant 5) against
float64 torch and against the plain tile-per-block kernel:

* whole-tile schedules (variant bit 16) run every tile's K-steps in the same order on the same MFMA sequence as the
  tile-per-block kernel -> results must be BIT-IDENTICAL to it;
* split schedules hand partial tiles from block to block (fixed fix-up order) -> float64-referenced tolerance, bit-identical
  from launch to launch, flag region left zeroed, no fault;
* every epilogue (linear / GRU z|r / GRU q), two sources, strides, ragged M / cout / channel tails, the batched correlation
  volume (batch folded into the tile index, supertile walk)."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PP = {"swz3_xcd": 53, "pad2_xcd": 52, "swz3": 51, "swz2_xcd": 50 + 1 + 2 + 8, "swz3_xcd_whole": 53 + 16, "pad2_whole": 50 + 16}
TILES = {"64x64_swz": 10, "64x128_swz": 11, "128x64_swz": 12, "128x128_swz": 13}


def pm(x):
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def _cases():
    rng = random.Random(7)
    out = [  # the update block's own launches at batch 1 / 2 (grid sizes on both sides of the stream-K thresholds)
        dict(B=1, H=55, W=128, kh=3, kw=3, real=[128], cout=512, stride=1, epi=0),
        dict(B=1, H=55, W=128, kh=1, kw=5, real=[384], cout=256, stride=1, epi=1),
        dict(B=2, H=47, W=156, kh=5, kw=1, real=[128, 256], cout=128, stride=1, epi=2),
        dict(B=1, H=55, W=128, kh=1, kw=1, real=[324], cout=256, stride=1, epi=0),
        dict(B=1, H=55, W=128, kh=3, kw=3, real=[256], cout=126, stride=1, epi=0),
        dict(B=3, H=46, W=62, kh=3, kw=3, real=[128], cout=64, stride=1, epi=0),
    ]
    for _ in range(10):
        k = rng.choice([(1, 1), (3, 3), (1, 5), (5, 1)])
        nsrc = rng.choice([1, 1, 2, 3])
        out.append(dict(B=rng.choice([1, 2, 3]), H=rng.randint(7, 40), W=rng.randint(9, 70), kh=k[0], kw=k[1],
                        real=[rng.choice([36, 64, 96, 126, 128, 146]) for _ in range(nsrc)], cout=rng.choice([40, 64, 126, 128, 192]),
                        stride=rng.choice([1, 1, 2]) if nsrc == 1 and k[0] == k[1] else 1, epi=0))
    return out


@pytest.mark.parametrize("c", _cases(), ids=lambda c: f"b{c['B']}_{c['H']}x{c['W']}_{c['kh']}x{c['kw']}s{c['stride']}_{'+'.join(map(str, c['real']))}to{c['cout']}e{c['epi']}")
def test_pp_kernel_matches_reference_and_tile_kernel(gpu, c):
    from ptlflow_amd.packing import pack_conv_weight
    ops = torch.ops.pfk
    torch.manual_seed(hash(str(c)) % 10000)
    B, H, W, kh, kw, cout, s, epi = c["B"], c["H"], c["W"], c["kh"], c["kw"], c["cout"], c["stride"], c["epi"]
    xs = [torch.randn(B, r, H, W, dtype=torch.float64) for r in c["real"]]
    buf = [(r + 3) // 4 * 4 for r in c["real"]]
    cin = sum(c["real"])
    w = torch.randn(cout, cin, kh, kw, dtype=torch.float64) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout, dtype=torch.float64) * 0.1
    pre = F.conv2d(torch.cat(xs, 1), w, b, stride=s, padding=(kh // 2, kw // 2))
    Ho, Wo = pre.shape[-2:]
    M = B * Ho * Wo
    srcs = [F.pad(pm(x.float()), (0, bf - r)).cuda() for x, r, bf in zip(xs, c["real"], buf)]
    segs, first = [], 0
    for r, bf in zip(c["real"], buf):
        segs.append((first, r, bf))
        first += r
    packed = pack_conv_weight(w.float(), segs).cuda()
    bias = b.float().cuda()
    Ch = cout // 2 if epi == 1 else cout
    h0 = torch.tanh(torch.randn(M, Ch, dtype=torch.float64))
    z0 = torch.rand(M, Ch, dtype=torch.float64)
    prem = pre.permute(0, 2, 3, 1).reshape(M, cout)
    if epi == 0:
        want = [F.relu(prem)]
    elif epi == 1:
        g = torch.sigmoid(prem)
        want = [g[:, :Ch], g[:, Ch:] * h0]
    else:
        want = [(1 - z0) * h0 + z0 * torch.tanh(prem)]
    ws = torch.zeros(ops.conv_workspace_bytes(), device=gpu, dtype=torch.uint8)

    def run(cfg):
        ops.debug_set_tile(cfg)
        try:
            out = torch.zeros(M, cout, device=gpu)
            hb, zb, rh = h0.float().cuda(), z0.float().cuda(), torch.zeros(M, Ch, device=gpu)
            if epi == 0:
                ops.conv2d(srcs, B, H, W, kh, kw, packed, bias, cout, 0, True, 1.0, out, None, None, None, ws, None, s, False)
                res = [out]
            elif epi == 1:
                ops.conv2d(srcs, B, H, W, kh, kw, packed, bias, cout, 1, False, 1.0, None, hb, zb, rh, ws)
                res = [zb, rh]
            else:
                ops.conv2d(srcs, B, H, W, kh, kw, packed, bias, cout, 2, False, 1.0, None, hb, zb, None, ws)
                res = [hb]
            torch.cuda.synchronize()
            return [r.clone() for r in res]
        finally:
            ops.debug_set_tile(-1)

    base = run(4)                                             # tile-per-block v3 kernel, padded LDS
    for name, cfg in PP.items():
        got = run(cfg)
        for g_, w_ in zip(got, want):
            err = float((g_.double().cpu() - w_).abs().max())
            assert err <= 3e-5 * (float(w_.abs().max()) + 1.0), f"{name}: err {err:.2e}"
        again = run(cfg)
        assert all(torch.equal(a, b_) for a, b_ in zip(got, again)), f"{name}: not deterministic from launch to launch"
        if name.endswith("whole"):
            assert all(torch.equal(a, b_) for a, b_ in zip(got, base)), f"{name}: differs from the tile-per-block kernel"
    for name, cfg in TILES.items():         # tile-per-block kernels: same K order per output element -> the same bits
        got = run(cfg)
        assert all(torch.equal(a, b_) for a, b_ in zip(got, base)), f"{name}: differs from the 64x64 tile kernel"
    off = ops.conv_workspace_fault_offset()
    assert int(ws[off: off + 4].view(torch.int32).item()) == 0
    assert bool((ws[ops.conv_workspace_bytes() - 768 * 64:][: 768 * 4] == 0).all()), "flag region not handed back zeroed"


@pytest.mark.parametrize("B,N1,N2,D", [(2, 7040, 7040, 256), (3, 1000, 777, 128), (1, 2852, 713, 256), (8, 330, 330, 64)])
def test_pp_kernel_correlation_volume(gpu, B, N1, N2, D):
    """K1 through the persistent kernel: batch folded into the tile index, 16 x 16 supertile walk on the big grids, whole tiles."""
    ops = torch.ops.pfk
    g = torch.Generator().manual_seed(N1 + N2)
    f1 = torch.randn(B, N1, D, generator=g).cuda()
    f2 = torch.randn(B, N2, D, generator=g).cuda()
    scale = 1.0 / math.sqrt(D)

    def run(cfg):
        ops.debug_set_tile(cfg)
        try:
            out = torch.empty(B, N1, N2, device=gpu)
            ops.corr_volume(f1, f2, scale, out)
            torch.cuda.synchronize()
            return out
        finally:
            ops.debug_set_tile(-1)

    base = run(10)
    ref = torch.bmm(f1.double(), f2.double().transpose(1, 2)) * scale
    assert float((base.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    for cfg in (53, 52, 51 + 8):
        assert torch.equal(run(cfg), base), f"cfg {cfg}"
