"""K8b — the update block with bf16 ACTIVATION STORAGE (`pfk_conv2d_b16` + the bf16-I/O forms of the lookup, the 7x7 flow convolution,
the flow head and the convex upsampling): what ptlflow/models/raft/update.py:6-153 computes under the reference's reduced-precision
switch (scripts/model_benchmark.py:317-319, validate.py:243-244 / torch.autocast(bfloat16)).

Oracle: the CPU restatement of the same layers (oracle/raft_oracle.py, F.conv2d in fp32) on the SAME bf16-rounded operands.  A bf16
product is exact in fp32, so the GPU result differs from it by the fp32 accumulation order only (gate 2e-5 (1 + |ref|), as for the
fp32 kernels) and, where the output is stored as bf16, by that one rounding (gate: half a bf16 ulp of the reference on top).  The
kernels with bf16 I/O around the GEMMs (lookup, 7x7, flow head, upsampling) are held BIT-EXACT to their fp32 forms: they run the same
fp32 arithmetic and only widen inputs / round outputs."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu

EPI_LINEAR, EPI_GRU_ZR, EPI_GRU_Q = 0, 1, 2
BF = torch.bfloat16


def pm(x):     # NCHW -> pixel-major [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def unpm(x, B, H, W):
    return x.reshape(B, H, W, -1).permute(0, 3, 1, 2).float().cpu()


def r16(x):    # round to bf16, keep fp32 storage
    return x.to(BF).float()


def pack16(w, segs):
    from ptlflow_amd.packing import pack_conv_weight
    return pack_conv_weight(w, segs, kpad=64).to(BF).cuda()


def close_b16(got, ref, out_bf16):
    """|got - ref| <= 2e-5 (1 + |ref|)  [+ half a bf16 ulp of ref when the output was rounded to bf16]"""
    tol = 2e-5 * (1 + ref.abs())
    if out_bf16:
        tol = tol + ref.abs() * 2.0 ** -8 + 1e-30
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} outside the gate, max err {(got - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 8, 9])
@pytest.mark.parametrize("B,H,W,segs,cout,kh,kw,relu,out_b16", [
    (1, 12, 20, [64], 96, 3, 3, True, True),
    (2, 9, 7, [328], 256, 1, 1, True, True),            # convc1: 324 correlation channels in a 328-wide bf16 row
    (1, 17, 33, [256], 126, 3, 3, True, True),          # encoder.conv: ragged cout, written into a channel slice
    (1, 10, 12, [256], 576, 1, 1, False, False),        # mask conv2 with an fp32 output
    (1, 11, 19, [40, 24], 40, 5, 1, False, True),       # two sources, channel tails inside a 64-channel K-step
    (1, 55, 128, [128], 64, 3, 3, True, True),          # one full 55x128 grid
])
def test_conv_b16_linear(gpu, cfg, B, H, W, segs, cout, kh, kw, relu, out_b16):
    torch.manual_seed(3)
    cin = sum(segs)
    x = r16(torch.randn(B, cin, H, W))
    wt = r16(torch.randn(cout, cin, kh, kw) / math.sqrt(cin * kh * kw))
    bias = torch.randn(cout)
    ref = F.conv2d(x, wt, bias, padding=(kh // 2, kw // 2))
    if relu:
        ref = F.relu(ref)
    ref = ref * 0.25
    M = B * H * W
    offs, o = [], 0
    srcs = []
    for c in segs:
        offs.append((o, c, c))
        # every source sits inside a wider row (ld > channels) at a channel offset, as the hx slices do
        buf = torch.full((M, c + 16), 7.0, device=gpu, dtype=BF)
        buf[:, 8:8 + c] = pm(x[:, o:o + c]).to(gpu, BF)
        srcs.append(buf[:, 8:8 + c])
        o += c
    obuf = torch.full((M, (cout + 7) // 8 * 8 + 16), 5.0, device=gpu, dtype=BF if out_b16 else torch.float32)
    out = obuf[:, 8:8 + cout]
    torch.ops.pfk.debug_set_b16(cfg)
    try:
        torch.ops.pfk.conv2d_b16(srcs, B, H, W, kh, kw, pack16(wt, offs), bias.cuda(), cout, EPI_LINEAR, relu, 0.25, out)
    finally:
        torch.ops.pfk.debug_set_b16(0)
    close_b16(unpm(out, B, H, W), ref, out_b16)
    assert bool((obuf[:, :8] == 5.0).all()) and bool((obuf[:, 8 + cout:] == 5.0).all()), "wrote outside its channel slice"


def test_conv_b16_padding_rows_are_zero_filled(gpu):
    """The convolution's zero padding is an out-of-range LDS-DMA lane: NaNs parked in the neighbouring rows of a source must not
    leak through a border tap (0 x NaN would), and rows past M / cout must not contaminate valid outputs."""
    torch.manual_seed(5)
    B, H, W, cin, cout = 1, 6, 9, 64, 64
    x = r16(torch.randn(B, cin, H, W))
    wt = r16(torch.randn(cout, cin, 3, 3) / 24)
    ref = F.conv2d(x, wt, None, padding=1)
    M = B * H * W
    big = torch.full((M + 2 * (W + 1), cin), float("nan"), device=gpu, dtype=BF)     # NaN guard rows around the map
    big[W + 1: W + 1 + M] = pm(x).to(gpu, BF)
    out = torch.zeros(M, cout, device=gpu, dtype=BF)
    torch.ops.pfk.conv2d_b16([big[W + 1: W + 1 + M]], B, H, W, 3, 3, pack16(wt, [(0, cin, cin)]), None, cout, EPI_LINEAR, False, 1.0, out)
    close_b16(unpm(out, B, H, W), ref, True)


@pytest.mark.parametrize("B,H,W,Ch,Cm,passes", [
    (1, 12, 16, 128, 128, ((1, 5, "1"), (5, 1, "2"))),   # SepConvGRU with the context slice hoisted (h | motion)
    (2, 10, 14, 96, 88, ((3, 3, ""),)),                  # ConvGRU widths of raft_small (82 motion channels in an 88-wide slice)
])
def test_gru_b16(gpu, B, H, W, Ch, Cm, passes):
    """z|r and q launches with the bf16 side buffers (z, r*h, the context term) against the oracle's GRU pass evaluated on the SAME
    rounded intermediates: z and r*h rounded to bf16 where the kernels store them, h kept in fp32."""
    torch.manual_seed(6)
    M = B * H * W
    h = torch.tanh(torch.randn(B, Ch, H, W))
    m = r16(torch.randn(B, Cm, H, W))
    ctx_zr = r16(torch.randn(B, 2 * Ch, H, W) * 0.3)
    ctx_q = r16(torch.randn(B, Ch, H, W) * 0.3)
    hx32 = pm(h).to(gpu)                                 # fp32 state
    hxb = torch.zeros(M, Ch + Cm, device=gpu, dtype=BF)  # [h | motion] bf16 twin
    hxb[:, :Ch] = hx32.to(BF)
    hxb[:, Ch:] = pm(m).to(gpu, BF)
    z = torch.zeros(M, Ch, device=gpu, dtype=BF)
    rh = torch.zeros(M, Ch, device=gpu, dtype=BF)
    href = h.clone()
    for kh, kw, sfx in passes:
        wz, wr, wq = (r16(torch.randn(Ch, Ch + Cm, kh, kw) / math.sqrt((Ch + Cm) * kh * kw)) for _ in range(3))
        segs = [(0, Ch, Ch), (Ch, Cm, Cm)]
        pad = (kh // 2, kw // 2)
        # reference on the rounded operands
        hb = r16(href)
        a = F.conv2d(torch.cat([hb, m], 1), torch.cat([wz, wr], 0), None, padding=pad) + ctx_zr
        zz, rr = torch.sigmoid(a[:, :Ch]), torch.sigmoid(a[:, Ch:])
        zz16, rh16 = r16(zz), r16(rr * hb)
        q = torch.tanh(F.conv2d(torch.cat([rh16, m], 1), wq, None, padding=pad) + ctx_q)
        href = (1 - zz16) * href + zz16 * q
        torch.ops.pfk.conv2d_b16([hxb[:, :Ch], hxb[:, Ch:]], B, H, W, kh, kw, pack16(torch.cat([wz, wr], 0), segs), None, 2 * Ch,
                                 EPI_GRU_ZR, False, 1.0, None, None, hxb[:, :Ch], z, rh, pm(ctx_zr).to(gpu, BF))
        # (the hardware exp / rcp of the gate epilogues are ~1 ulp each: far inside half a bf16 ulp)
        assert (unpm(z, B, H, W) - zz).abs().max() <= 2.0 ** -8
        assert (unpm(rh, B, H, W) - rr * hb).abs().max() <= 2.0 ** -8
        torch.ops.pfk.conv2d_b16([rh, hxb[:, Ch:]], B, H, W, kh, kw, pack16(wq, segs), None, Ch, EPI_GRU_Q, False, 1.0, None,
                                 hx32, hxb[:, :Ch], z, None, pm(ctx_q).to(gpu, BF))
    got = unpm(hx32, B, H, W)
    # z / r*h may round to the neighbouring bf16 value (1 ulp = 2^-8 relative) where the GPU's accumulation order lands on the other
    # side of a rounding boundary: that moves h by <= 2^-8 |q - h| per pass
    assert (got - href).abs().max() <= 1.2e-2
    assert (got - href).abs().mean() <= 3e-4
    assert torch.equal(hxb[:, :Ch].float().cpu(), hx32.to(BF).float().cpu()), "the bf16 twin of h is not its rounding"


def test_lookup_bf16_output_is_the_rounded_fp32_lookup(gpu):
    from ptlflow_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(9)
    B, D, h, w = 2, 64, 24, 40
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    c = O.coords_grid(B, h, w) + torch.rand(B, 2, h, w, generator=g) * 8 - 4
    for vt in (torch.float32, BF):
        for layout in ("blocked", "rowmajor"):
            cb = CorrBlock(f1.to(gpu), f2.to(gpu), 4, 4, volume_dtype=vt, layout=layout)
            ref = cb.lookup_pm(c.to(gpu)).clone()                               # fp32 [M, >= 324]
            out = torch.full((B * h * w, 328), 3.0, device=gpu, dtype=BF)
            cb.lookup_pm(c.to(gpu), out=out)
            assert torch.equal(out[:, :324], ref[:, :324].to(BF)), (vt, layout)
            assert bool((out[:, 324:] == 3.0).all()), "wrote the pad columns"


def test_cin2_and_flow_head_b16_io(gpu):
    from ptlflow_amd.packing import pack_cin2_weight, pack_flow_head_weight
    torch.manual_seed(8)
    B, H, W = 2, 13, 37
    M = B * H * W
    flow = (torch.randn(M, 4) * 3).to(gpu)
    wt = pack_cin2_weight(torch.randn(128, 2, 7, 7) * 0.1).to(gpu)
    bias = torch.randn(128).to(gpu)
    o32 = torch.empty(M, 128, device=gpu)
    o16 = torch.empty(M, 128, device=gpu, dtype=BF)
    torch.ops.pfk.conv_cin2(flow, wt, bias, o32, B, H, W, 7, True)
    torch.ops.pfk.conv_cin2(flow, wt, bias, o16, B, H, W, 7, True)
    assert torch.equal(o16, o32.to(BF))
    # flow head conv2 + coordinate update: bf16 input == the fp32 kernel on the widened input, plus the bf16 copy of the flow
    x16 = torch.randn(M, 256, device=gpu).to(BF)
    w2 = pack_flow_head_weight(torch.randn(2, 256, 3, 3) * 0.05).to(gpu)
    b2 = torch.randn(2).to(gpu)
    c0 = O.coords_grid(B, H, W).to(gpu)
    ca, cb = c0.clone() + 1.5, c0.clone() + 1.5
    da, db = torch.empty_like(c0), torch.empty_like(c0)
    fa, fb = torch.zeros(M, 4, device=gpu), torch.zeros(M, 4, device=gpu)
    f16 = torch.zeros(M, 8, device=gpu, dtype=BF)
    torch.ops.pfk.flow_delta(x16.float(), w2, b2, c0, ca, da, fa[:, :2])
    torch.ops.pfk.flow_delta(x16, w2, b2, c0, cb, db, fb[:, :2], f16[:, 6:8])
    assert torch.equal(ca, cb) and torch.equal(da, db) and torch.equal(fa, fb)
    assert torch.equal(f16[:, 6:8], fb[:, :2].to(BF)) and bool((f16[:, :6] == 0).all())


def test_convex_upsample_bf16_mask(gpu):
    torch.manual_seed(11)
    B, H, W = 2, 9, 21
    M = B * H * W
    mask16 = (torch.randn(M, 576, device=gpu) * 2).to(BF)
    flow = torch.randn(M, 4, device=gpu) * 4
    a = torch.empty(B, 2, 8 * H, 8 * W, device=gpu)
    b = torch.empty_like(a)
    torch.ops.pfk.convex_upsample_pm(flow[:, :2], mask16.float(), a)
    torch.ops.pfk.convex_upsample_pm(flow[:, :2], mask16, b)
    assert torch.equal(a, b)


@pytest.mark.parametrize("small", [False, True])
def test_engine_b16_step_against_the_oracle_block(gpu, small):
    """One update-block call of the K8b engine (lookup result in, net / delta / mask out) against the oracle's block evaluated in fp32
    on the same inputs: the distance is the bf16 operand rounding (a few 1e-3 relative on O(1) activations), nothing structural."""
    from ptlflow_amd.raft import RAFT
    from ptlflow_amd.update import UpdateEngine
    torch.manual_seed(12)
    model = RAFT(small=small).load_synthetic(4)
    P = {k[len("update_block."):]: v for k, v in model.state_dict().items() if k.startswith("update_block.")}
    s = model.spec
    B, H, W = 1, 16, 24
    net = torch.tanh(torch.randn(B, s.hidden, H, W))
    inp = torch.relu(torch.randn(B, s.context, H, W))
    corr = torch.randn(B, s.corr_channels, H, W)
    flow = torch.randn(B, 2, H, W) * 2
    ref = (O.small_update_block if small else O.basic_update_block)(P, net, inp, corr, flow)
    eng = UpdateEngine({k: v for k, v in P.items()}, s, gpu, "bf16")
    assert eng.b16
    eng.bind(B, H, W)
    eng.load_state(net.to(gpu), inp.to(gpu))
    torch.ops.pfk.nchw_to_pm(flow.to(gpu).contiguous(), eng.flow_view)
    eng.flow_changed()
    eng.corr16[:, : s.corr_channels] = pm(corr).to(gpu, BF)
    c0 = O.coords_grid(B, H, W).to(gpu)
    c1 = c0 + flow.to(gpu)
    delta = torch.empty_like(c0)
    eng.motion_and_gru(eng.corr16)
    eng.heads(c0, c1, delta, want_mask=True)
    rnet, rmask, rdelta = ref
    assert (eng.net_nchw().cpu() - rnet).abs().max() < 6e-2 and (eng.net_nchw().cpu() - rnet).abs().mean() < 6e-3
    assert (delta.cpu() - rdelta).abs().max() < 8e-2 * (1 + rdelta.abs().max())
    if rmask is not None:
        assert (eng.mask_nchw().cpu() - rmask).abs().mean() < 2e-2 * (1 + rmask.abs().mean())


@pytest.mark.parametrize("Bt,N,C", [(2, 920, 128), (3, 333, 64)])
def test_conv_b16_batched_gemm_with_bf16_residual(gpu, Bt, N, C):
    """GMA's `fmap + gamma * (attn @ v)` (gma/gma_utils.py:100-113) as ONE batched K8b launch: per pair A = attn[b] [N][N] bf16 (rows
    padded to 16 bytes, pad columns zero), B = v[b]^T [C][K], bf16 residual rows and bf16 output inside a wider buffer."""
    torch.manual_seed(14)
    Np, K = (N + 7) // 8 * 8, ((N + 7) // 8 * 8 + 63) // 64 * 64
    attn = torch.zeros(Bt, N, Np)
    attn[:, :, :N] = torch.softmax(torch.randn(Bt, N, N) * 2, dim=-1)
    attn = r16(attn)
    v = r16(torch.randn(Bt, N, C))
    mf = r16(torch.randn(Bt, N, C))
    gamma = 0.37
    ref = mf + gamma * torch.bmm(attn[:, :, :N], v)
    vT = torch.zeros(Bt, C, K, device=gpu, dtype=BF)
    vT[:, :, :N] = v.transpose(1, 2).to(gpu, BF)
    buf = torch.full((Bt, N, 2 * C + 8), 3.0, device=gpu, dtype=BF)
    buf[:, :, :C] = mf.to(gpu, BF)
    hw = next((h, N // h) for h in range(int(math.sqrt(N)), 0, -1) if N % h == 0)
    torch.ops.pfk.conv2d_b16([attn.to(gpu, BF)], 1, hw[0], hw[1], 1, 1, vT, None, C, EPI_LINEAR, False, gamma, buf[:, :, C: 2 * C],
                             None, None, None, None, buf[:, :, :C])
    close_b16(buf[:, :, C: 2 * C].float().cpu(), ref, True)
    assert bool((buf[:, :, 2 * C:] == 3.0).all()) and torch.equal(buf[:, :, :C].float().cpu(), mf)


def test_engine_b16_gma_step(gpu):
    """One GMA update-block call on the K8b engine (aggregate branch on the bf16 attention map) against the oracle's block in fp32."""
    from ptlflow_amd.raft import GMA
    from ptlflow_amd.update import UpdateEngine
    torch.manual_seed(15)
    model = GMA().load_synthetic(5)
    with torch.no_grad():
        model.update_block.aggregator.gamma.fill_(0.4)
    P = {k[len("update_block."):]: v for k, v in model.state_dict().items() if k.startswith("update_block.")}
    s = model.spec
    B, H, W = 2, 16, 24
    N = H * W
    net = torch.tanh(torch.randn(B, s.hidden, H, W))
    inp = torch.relu(torch.randn(B, s.context, H, W))
    corr = torch.randn(B, s.corr_channels, H, W)
    flow = torch.randn(B, 2, H, W) * 2
    attn = torch.softmax(torch.randn(B, 1, N, N), dim=-1)
    rnet, rmask, rdelta = O.gma_update_block(P, net, inp, corr, flow, attn)
    eng = UpdateEngine(P, s, gpu, "bf16")
    assert eng.b16
    eng.bind(B, H, W)
    eng.load_state(net.to(gpu), inp.to(gpu))
    eng.set_attention(attn.to(gpu))
    torch.ops.pfk.nchw_to_pm(flow.to(gpu).contiguous(), eng.flow_view)
    eng.flow_changed()
    eng.corr16[:, : s.corr_channels] = pm(corr).to(gpu, BF)
    c0 = O.coords_grid(B, H, W).to(gpu)
    delta = torch.empty_like(c0)
    eng.motion_and_gru(eng.corr16)
    eng.heads(c0, c0 + flow.to(gpu), delta, want_mask=True)
    assert (eng.net_nchw().cpu() - rnet).abs().max() < 6e-2 and (eng.net_nchw().cpu() - rnet).abs().mean() < 6e-3
    assert (delta.cpu() - rdelta).abs().max() < 8e-2 * (1 + rdelta.abs().max())


@pytest.mark.parametrize("cin,cout,k,stride,H,W", [(64, 96, 3, 2, 21, 34), (64, 96, 1, 2, 20, 33), (96, 128, 3, 2, 11, 16)])
def test_conv_b16_strided(gpu, cin, cout, k, stride, H, W):
    """The encoders' stride-2 3x3 / 1x1 convolutions (raft/extractor.py:31-34, 50-58) on the K8b kernel, residual add + second relu
    fused (the ResidualBlock's `relu(x + y)`), bf16 residual rows."""
    torch.manual_seed(16)
    B = 2
    x = r16(torch.randn(B, cin, H, W))
    wt = r16(torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k))
    bias = torch.randn(cout)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = r16(torch.randn(B, cout, Ho, Wo))
    ref = F.relu(res + F.relu(F.conv2d(x, wt, bias, stride=stride, padding=k // 2)))
    out = torch.zeros(B * Ho * Wo, cout, device=gpu, dtype=BF)
    torch.ops.pfk.conv2d_b16([pm(x).to(gpu, BF)], B, H, W, k, k, pack16(wt, [(0, cin, cin)]), bias.cuda(), cout, EPI_LINEAR, True, 1.0, out,
                             None, None, None, None, pm(res).to(gpu, BF), stride, True)
    close_b16(unpm(out, B, Ho, Wo), ref, True)


@pytest.mark.parametrize("norm,small", [("instance", False), ("batch", False), ("instance", True), ("none", True)])
def test_encoder_b16_against_the_oracle_encoder(gpu, norm, small):
    """BasicEncoder / SmallEncoder (raft/extractor.py:122-267) on the K8b path (bf16 activation storage between the layers, fp32
    instance-norm statistics) against the oracle's fp32 encoder: the distance is the bf16 rounding of ~20 layers of activations."""
    from ptlflow_amd.encoder import EncoderEngine
    from ptlflow_amd.raft import Encoder
    torch.manual_seed(17)
    out_dim = 128 if small else 256
    enc = Encoder(out_dim, norm, small).eval()
    if norm == "batch":
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    P = enc.state_dict()
    x = torch.rand(2, 3, 96, 136) * 2 - 1
    ref = O.encoder({k: v.clone() for k, v in P.items()}, x, norm, small)
    got = EncoderEngine(P, norm, gpu, "bf16", small)(x.to(gpu)).float().cpu()
    f32 = EncoderEngine(P, norm, gpu, "fp32", small)(x.to(gpu)).float().cpu()
    assert got.shape == ref.shape
    scale = ref.abs().mean()
    assert (f32 - ref).abs().max() < 1e-3 * (1 + ref.abs().max())
    # the yardstick: the same encoder under the reference's reduced-precision switch on the CPU (torch.autocast(bfloat16): 16-bit
    # convolution outputs, fp32 instance norms on them) — this path may not be further from fp32 than 1.5x that, or 2 % of the scale
    with torch.autocast("cpu", dtype=torch.bfloat16):
        auto = O.encoder({k: v.clone() for k, v in P.items()}, x, norm, small).float()
    gap = (auto - ref).abs().mean()
    err = (got - ref).abs().mean()
    assert err < max(2e-2 * scale, 1.5 * gap), f"mean err {err:.3e}, autocast gap {gap:.3e}, scale {scale:.3e}"
    assert (got - ref).abs().max() < 0.25 * (1 + ref.abs().max())


@pytest.mark.parametrize("B,H,W", [(1, 55, 128), (2, 13, 17), (1, 8, 9), (3, 47, 156)])
def test_mask_upsample_fused_b16(gpu, B, H, W):
    """K13b `pfk_mask_upsample_b16` (mask conv2 + softmax + convex upsampling on bf16 operands, no mask in memory) against the two K8b
    launches it replaces (`conv2d_b16` with scale 0.25 -> bf16 [M, 576] logits -> `convex_upsample_pm`): bit-identical (every logit is
    rounded to bf16 where the unfused launch stores it); and against the oracle's `convex_upsample` on the fp32 convolution of the
    bf16-rounded operands, to the distance one bf16 rounding of the logits makes."""
    from ptlflow_amd.packing import pack_conv_weight, permute_mask_head
    torch.manual_seed(19)
    M, cin = B * H * W, 256
    fm = r16(torch.randn(M, 512))                  # fh | mask hidden: the kernel reads the second half as a strided view
    wt = r16(torch.randn(576, cin, 1, 1) / math.sqrt(cin))
    bias = torch.randn(576) * 0.1
    hx = torch.randn(M, 8)
    wp, bp = permute_mask_head(pack_conv_weight(wt, [(0, cin, cin)]), bias)
    fm_g, hx_g = fm.to(gpu, BF), hx.cuda()
    x_g, flow_g = fm_g[:, 256:], hx_g[:, 4:6]
    mask = torch.empty(M, 576, device=gpu, dtype=BF)
    torch.ops.pfk.conv2d_b16([x_g], B, H, W, 1, 1, pack16(wt, [(0, cin, cin)]), bias.cuda(), 576, EPI_LINEAR, False, 0.25, mask)
    want = torch.empty(B, 2, 8 * H, 8 * W, device=gpu)
    torch.ops.pfk.convex_upsample_pm(flow_g, mask, want)
    got = torch.full((B, 2, 8 * H, 8 * W), 7.0, device=gpu)
    torch.ops.pfk.mask_upsample(x_g, wp.to(gpu, BF), bp.cuda(), 0.25, flow_g, got)
    assert torch.equal(got, want), f"max diff {(got - want).abs().max().item():.3e}"
    x_nchw = fm[:, 256:].reshape(B, H, W, cin).permute(0, 3, 1, 2)
    mask_ref = r16(0.25 * F.conv2d(x_nchw, wt, bias))
    flow_ref = hx[:, 4:6].reshape(B, H, W, 2).permute(0, 3, 1, 2)
    ref = O.convex_upsample(flow_ref, mask_ref)
    assert (got.cpu() - ref).abs().max() < 2e-2 * (1 + ref.abs().max())


@pytest.mark.parametrize("kind,B,H,W", [("raft", 8, 480, 640), ("raft", 1, 184, 320), ("gma", 2, 200, 328)])
def test_fused_mask_upsample_b16_is_bit_identical_in_the_forward(gpu, kind, B, H, W):
    """`fuse_mask_upsample` on the K8b path: whole forwards with K13b and with the pair (both on the side stream where it opens) give
    the same bits, over repeated forwards on fresh inputs."""
    from ptlflow_amd.raft import GMA, RAFT
    make = (lambda: GMA(iters=5, conv_precision="bf16")) if kind == "gma" else (lambda: RAFT(iters=5, conv_precision="bf16"))
    a, b = make().load_synthetic(5).eval().cuda(), make().load_synthetic(5).eval().cuda()
    a.fuse_mask_upsample, b.fuse_mask_upsample = False, True
    for seed in (1, 2, 1):
        x = O.smooth_pair(B, H, W, seed=seed).cuda()
        fa, fb = a({"images": x}), b({"images": x})
        assert torch.equal(fa["flows"], fb["flows"]) and torch.equal(fa["flow_small"], fb["flow_small"]), seed
