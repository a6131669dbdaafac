"""Randomised shapes for the implicit-GEMM convolution family: forward (fp32 and split-bf16 x6), strided, multi-source
with zero-padded channel tails, residual / relu epilogues, and the weight gradient — each against torch in float64.
Seeded: the same 40 cases every run."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def pm(x):
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        k = rng.choice([(1, 1), (3, 3), (1, 5), (5, 1), (3, 3), (7, 7)])
        nsrc = rng.choice([1, 1, 2, 3]) if k != (7, 7) else 1
        real = [rng.choice([4, 36, 64, 96, 126, 128, 146, 324]) if k != (7, 7) else rng.choice([2, 3, 4]) for _ in range(nsrc)]
        out.append(dict(B=rng.choice([1, 2, 3]), H=rng.randint(5, 23), W=rng.randint(5, 37), kh=k[0], kw=k[1], real=real,
                        buf=[(r + 3) // 4 * 4 for r in real], cout=rng.choice([2, 40, 64, 126, 128, 192, 256]),
                        relu=rng.random() < 0.5, residual=rng.random() < 0.3,
                        stride=rng.choice([1, 1, 1, 2]) if nsrc == 1 and k[0] == k[1] else 1,
                        seed=rng.randint(0, 10 ** 6)))
    return out


@pytest.mark.parametrize("c", cases(40, 2024), ids=lambda c: f"{c['kh']}x{c['kw']}s{c['stride']}_{'+'.join(map(str, c['real']))}to{c['cout']}")
def test_conv_family_random_shapes(gpu, c):
    from ptlflow_amd.packing import pack_conv_weight, split_bf16_planes
    ops = torch.ops.pfk
    torch.manual_seed(c["seed"])
    B, H, W, kh, kw, cout, s = c["B"], c["H"], c["W"], c["kh"], c["kw"], c["cout"], c["stride"]
    xs = [torch.randn(B, r, H, W, dtype=torch.float64) for r in c["real"]]
    cin = sum(c["real"])
    w = torch.randn(cout, cin, kh, kw, dtype=torch.float64) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout, dtype=torch.float64) * 0.1
    ref = F.conv2d(torch.cat(xs, 1), w, b, stride=s, padding=(kh // 2, kw // 2))
    Ho, Wo = ref.shape[-2:]
    if c["relu"]:
        ref = F.relu(ref)
    res = torch.randn(B, cout, Ho, Wo, dtype=torch.float64) if c["residual"] else None
    if res is not None:
        ref = F.relu(res + ref)
    srcs = [F.pad(pm(x.float()), (0, bf - r)).cuda() for x, r, bf in zip(xs, c["real"], c["buf"])]
    segs, first = [], 0
    for r, bf in zip(c["real"], c["buf"]):
        segs.append((first, r, bf))
        first += r
    packed = pack_conv_weight(w.float(), segs).cuda()
    scale = float(ref.abs().max()) + 1e-6
    for weight, tol in ((packed, 3e-5), (split_bf16_planes(packed, 3), 3e-5)):
        out = torch.zeros(B * Ho * Wo, cout, device=gpu)
        ops.conv2d(srcs, B, H, W, kh, kw, weight, b.float().cuda(), cout, 0, c["relu"], 1.0, out, None, None, None, None,
                   None if res is None else pm(res.float()).cuda(), s, res is not None)
        got = out.view(B, Ho, Wo, cout).permute(0, 3, 1, 2).double().cpu()
        err = float((got - ref).abs().max())
        assert err <= tol * scale + tol, f"{'split' if weight.dtype == torch.bfloat16 else 'fp32'}: err {err:.3e} scale {scale:.3e}"
    if s == 1 and cout % 4 == 0:
        # weight gradient of the same convolution for a random upstream gradient
        gy = torch.randn(B, cout, H, W, dtype=torch.float64)
        xcat = torch.cat(xs, 1).requires_grad_(False)
        wref = torch.nn.grad.conv2d_weight(xcat, w.shape, gy, padding=(kh // 2, kw // 2))
        ktot = sum(kh * kw * ((bf + 31) // 32 * 32) for bf in c["buf"])
        pk = torch.empty(cout, ktot, device=gpu)
        ops.conv_wgrad(srcs, pm(gy.float()).cuda(), B, H, W, kh, kw, pk)
        from ptlflow_amd.train import _Geometry, _unpack_wgrad
        got = _unpack_wgrad(pk, tuple(w.shape), segs, _Geometry(B, H, W, kh, kw)).double().cpu()
        wscale = float(wref.abs().max()) + 1e-6
        assert float((got - wref).abs().max()) <= 1e-4 * wscale, "weight gradient"
