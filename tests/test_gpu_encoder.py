"""Parity of the encoder kernels against the CPU oracle / torch fp64 on the same inputs.

Tolerance: the fp32 kernels' |err| <= 2e-5 * (1 + |ref|) for single ops; the whole BasicEncoder (8 normalised layers
deep) is held to 2e-4 relative to the output's scale, and the model-level gate stays the EPE gate of test_gpu_model.py
(which now runs through these kernels by default)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


def close(a, b, rtol=2e-5, atol=2e-5):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"


def pm(x):
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda()


def unpm(x, B, H, W):
    return x.view(B, H, W, -1).permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("kernel", ["mfma", "valu", "mfma_bf16_out"])
@pytest.mark.parametrize("B,H,W,cout,relu", [(1, 37, 70, 64, True), (2, 24, 64, 64, False), (1, 16, 130, 32, True), (3, 65, 131, 64, True),
                                             (1, 7, 5, 32, False), (1, 40, 48, 48, True)])
def test_stem(gpu, B, H, W, cout, relu, kernel):
    """Conv2d(3, cout, 7, stride 2, padding 3) from the NCHW image (raft/extractor.py:146): the MFMA implicit-GEMM kernel (cout 32 / 64,
    fp32 and bf16 output rows), the VALU kernel (every other width; forced here for 32 / 64 too) against float64 torch."""
    if kernel == "mfma_bf16_out" and cout not in (32, 64):
        pytest.skip("the bf16-output stem exists for the encoders' widths only")
    torch.manual_seed(1)
    img = torch.randn(B, 3, H, W)
    wt = torch.randn(cout, 3, 7, 7) / math.sqrt(147)
    bias = torch.randn(cout) * 0.1
    ref = F.conv2d(img.double(), wt.double(), bias.double(), stride=2, padding=3).float()
    if relu:
        ref = F.relu(ref)
    Ho, Wo = ref.shape[-2:]
    b16 = kernel == "mfma_bf16_out"
    out = torch.full((B * Ho * Wo, cout + 4), 7.0, device=gpu, dtype=torch.bfloat16 if b16 else torch.float32)
    w = wt.permute(2, 3, 1, 0).reshape(49, 3, cout).contiguous().cuda()
    torch.ops.pfk.debug_set_stem_valu(1 if kernel == "valu" else 0)
    try:
        torch.ops.pfk.conv_stem(img.cuda(), w, bias.cuda(), out[:, :cout], relu)
    finally:
        torch.ops.pfk.debug_set_stem_valu(0)
    if b16:   # one rounding of the fp32 result
        close(unpm(out[:, :cout], B, Ho, Wo), ref, rtol=2 ** -8, atol=2e-5)
    else:
        close(unpm(out[:, :cout], B, Ho, Wo), ref)
    assert bool((out[:, cout:] == 7.0).all())


@pytest.mark.parametrize("nsplit", [0, 3])
@pytest.mark.parametrize("B,H,W,cin,cout,k,stride", [
    (1, 22, 36, 64, 96, 3, 2), (2, 17, 33, 96, 128, 3, 2), (1, 17, 33, 64, 96, 1, 2), (1, 110, 256, 64, 96, 3, 2),
])
def test_strided_conv(gpu, nsplit, B, H, W, cin, cout, k, stride):
    from ptlflow_amd.packing import pack_conv_weight, split_bf16_planes
    torch.manual_seed(2)
    x = torch.randn(B, cin, H, W)
    wt = torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)
    bias = torch.randn(cout) * 0.1
    ref = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), stride=stride, padding=k // 2)).float()
    Ho, Wo = ref.shape[-2:]
    res = torch.randn(B, cout, Ho, Wo)
    ref = F.relu(res + ref)
    packed = pack_conv_weight(wt, [(0, cin, cin)])
    w = (packed if nsplit == 0 else split_bf16_planes(packed, nsplit)).cuda()
    out = torch.zeros(B * Ho * Wo, cout, device=gpu)
    torch.ops.pfk.conv2d([pm(x)], B, H, W, k, k, w, bias.cuda(), cout, 0, True, 1.0, out, None, None, None, None, pm(res),
                         stride, True)
    close(unpm(out, B, Ho, Wo), ref)


@pytest.mark.parametrize("B,H,W,C", [(2, 20, 33, 64), (3, 9, 14, 96), (1, 55, 128, 128)])
def test_instance_norm(gpu, B, H, W, C):
    torch.manual_seed(3)
    x = torch.randn(B, C, H, W) * 2.0 + 0.7
    res = torch.randn(B, C, H, W)
    ref1 = F.relu(F.instance_norm(x, eps=1e-5))
    ref2 = F.relu(res + F.relu(F.instance_norm(x, eps=1e-5)))
    ops = torch.ops.pfk
    ws = torch.empty(ops.instnorm_workspace_bytes(B, C), device=gpu, dtype=torch.uint8)
    mean = torch.empty(B * C, device=gpu)
    rstd = torch.empty(B * C, device=gpu)
    xp = pm(x)
    ops.instnorm_stats(xp, B, H * W, 1e-5, mean, rstd, ws)
    close(mean.view(B, C), x.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)
    close(rstd.view(B, C), 1.0 / torch.sqrt(x.var(dim=(2, 3), unbiased=False) + 1e-5), rtol=1e-5, atol=1e-6)
    out = torch.empty_like(xp)
    ops.norm_apply(xp, mean, rstd, None, out, B, H * W, True, False)
    close(unpm(out, B, H, W), ref1, rtol=1e-5, atol=1e-5)
    ops.norm_apply(xp, mean, rstd, pm(res), xp, B, H * W, True, True)     # in place, with the residual
    close(unpm(xp, B, H, W), ref2, rtol=1e-5, atol=1e-5)


def _encoder_params(kind, out_dim, seed, small=False):
    from ptlflow_amd.raft import Encoder
    from ptlflow_amd.synth import synth_state_dict
    enc = Encoder(out_dim, kind, small)
    own = enc.state_dict()
    shapes = {"fnet." + k: tuple(v.shape) for k, v in own.items()}      # "fnet." prefix: the encoder init statistics
    return {k[len("fnet."):]: v for k, v in synth_state_dict(shapes, seed).items()}


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16x6", 2e-4), ("bf16x3", 2e-3)])
@pytest.mark.parametrize("kind,B,H,W", [("instance", 2, 64, 96), ("batch", 1, 72, 136), ("instance", 1, 136, 72)])
def test_basic_encoder(gpu, kind, B, H, W, precision, tol):
    """Whole BasicEncoder (extractor.py:172-194) vs the oracle; 136 -> 68 -> 34 -> 17 exercises odd sizes."""
    from ptlflow_amd.encoder import EncoderEngine
    P = _encoder_params(kind, 256, seed=11)
    x = O.smooth_pair(B, H, W, seed=5)[:, 0]
    x = (x - 0.5) * 2.0
    ref = O.encoder(P, x, kind)
    eng = EncoderEngine(P, kind, gpu, precision)
    out = eng(x.cuda())
    assert tuple(out.shape) == tuple(ref.shape)
    scale = float(ref.abs().max())
    close(out, ref, rtol=tol, atol=tol * scale)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16x6", 2e-4)])
@pytest.mark.parametrize("kind,out_dim,B,H,W", [("instance", 128, 2, 64, 96), ("none", 160, 1, 72, 136), ("batch", 128, 1, 136, 72)])
def test_small_encoder(gpu, kind, out_dim, B, H, W, precision, tol):
    """Whole SmallEncoder of raft_small (extractor.py:197-267: bottleneck blocks 1x1 -> 3x3(stride) -> 1x1 at widths 8 / 16 / 24,
    instance norm for fnet, no norm for cnet) vs the oracle."""
    from ptlflow_amd.encoder import EncoderEngine
    P = _encoder_params(kind, out_dim, seed=12, small=True)
    x = (O.smooth_pair(B, H, W, seed=5)[:, 0] - 0.5) * 2.0
    ref = O.encoder(P, x, kind, small=True)
    out = EncoderEngine(P, kind, gpu, precision, small=True)(x.cuda())
    assert tuple(out.shape) == tuple(ref.shape)
    close(out, ref, rtol=tol, atol=tol * float(ref.abs().max()))


def test_encoder_batch_chunking(gpu):
    """Batches whose activation matrices would exceed the kernels' 32-bit byte offsets are processed in chunks."""
    from ptlflow_amd.encoder import EncoderEngine
    P = _encoder_params("instance", 256, seed=13)
    x = ((O.smooth_pair(3, 64, 96, seed=6)[:, 0]) - 0.5) * 2.0
    eng = EncoderEngine(P, "instance", gpu)
    whole = eng(x.cuda())
    eng.max_matrix_bytes = 32 * 48 * 128 * 4          # one image per chunk
    assert torch.equal(eng(x.cuda()), whole)


def test_raft_torch_encoders_still_work(gpu):
    """native_encoders=False keeps the torch (MIOpen) encoders: same flow within the EPE gate."""
    from ptlflow_amd.raft import RAFT
    x = O.smooth_pair(1, 128, 192, seed=3).cuda()
    a = RAFT(iters=4).load_synthetic(7).eval().cuda()
    b = RAFT(iters=4, native_encoders=False).load_synthetic(7).eval().cuda()
    fa, fb = a({"images": x})["flows"], b({"images": x})["flows"]
    mean, mx = O.epe(fa[:, 0].cpu(), fb[:, 0].cpu())
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


@pytest.mark.parametrize("kind", ["instance", "batch"])
def test_pfk_encoder_wrapper_contract(gpu, kind):
    """Seam B4: PfkEncoder around a BasicEncoder-shaped module — list-of-two-images contract (extractor.py:172-193),
    state_dict keys unchanged, eval GPU inference on the kernels, training mode deferred to the wrapped module."""
    from ptlflow_amd.encoder import PfkEncoder
    from ptlflow_amd.raft import Encoder
    ref = Encoder(256, kind, False)
    ref.load_state_dict(_encoder_params(kind, 256, seed=21))
    ref = ref.cuda().eval()
    wrapped = PfkEncoder(ref, "fp32")
    assert set(wrapped.state_dict()) == set(ref.state_dict())
    x = (O.smooth_pair(2, 64, 96, seed=8) - 0.5) * 2.0
    P = {k: v.cpu() for k, v in ref.state_dict().items()}
    want1, want2 = O.encoder(P, x[:, 0], kind), O.encoder(P, x[:, 1], kind)
    with torch.no_grad():
        f1, f2 = wrapped([x[:, 0].cuda(), x[:, 1].cuda()])
    scale = float(want1.abs().max())
    if kind == "instance":      # per-image statistics: batching the two frames changes nothing
        close(f1, want1, rtol=2e-4, atol=2e-4 * scale)
        close(f2, want2, rtol=2e-4, atol=2e-4 * scale)
    else:
        close(f1, want1, rtol=2e-4, atol=2e-4 * scale)
    assert wrapped._engine is not None
    wrapped.train()
    out = wrapped(x[:, 0].cuda())         # training mode: the wrapped module's own forward (autograd graph intact)
    assert out.requires_grad
