"""Training path of the update block: forward values and every gradient of the libpfk composition
(ptlflow_amd/train.py: conv forward / dgrad / wgrad on the MFMA implicit-GEMM kernel) against torch autograd in float64 on
the CPU — the oracle's functional update block, which is the reference's arithmetic (raft/update.py:6-153).

Tolerance: 2e-4 of each tensor's scale (fp32 sums over up to ~10^4 pixels in the weight gradients)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


def rel_close(a, b, tol=2e-4, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("B,H,W,cins,bufs,cout,kh,kw,relu", [
    (2, 11, 17, (64,), (64,), 96, 3, 3, True),
    (1, 12, 20, (96, 146), (96, 148), 128, 1, 5, False),       # two sources, the second with 2 zero-pad channels
    (1, 12, 20, (128, 256), (128, 256), 128, 5, 1, False),
    (2, 9, 13, (256,), (256,), 2, 3, 3, False),                # flow head conv2: cout = 2
    (1, 10, 14, (2,), (4,), 64, 7, 7, True),                   # convf1: 2 real channels in a 4-channel buffer
    (1, 14, 18, (324,), (324,), 256, 1, 1, True),
])
def test_conv_pm_forward_and_gradients(gpu, B, H, W, cins, bufs, cout, kh, kw, relu):
    from ptlflow_amd.train import conv_pm
    torch.manual_seed(7)
    M = B * H * W
    xs = [torch.randn(B, c, H, W, dtype=torch.float64, requires_grad=True) for c in cins]
    w = (torch.randn(cout, sum(cins), kh, kw, dtype=torch.float64) / math.sqrt(sum(cins) * kh * kw)).requires_grad_()
    b = (torch.randn(cout, dtype=torch.float64) * 0.1).requires_grad_()
    ref = F.conv2d(torch.cat(xs, 1), w, b, padding=(kh // 2, kw // 2))
    if relu:
        ref = F.relu(ref)
    gout = torch.randn_like(ref)
    ref.backward(gout)

    srcs = []
    for x, c, cb in zip(xs, cins, bufs):
        pm = x.detach().float().permute(0, 2, 3, 1).reshape(M, c)
        pm = F.pad(pm, (0, cb - c)).cuda().requires_grad_()
        srcs.append(pm)
    wg = w.detach().float().cuda().requires_grad_()
    bg = b.detach().float().cuda().requires_grad_()
    out = conv_pm(srcs, wg, bg, B, H, W, relu, list(cins))
    out.backward(gout.float().permute(0, 2, 3, 1).reshape(M, cout).cuda())
    rel_close(out.view(B, H, W, cout).permute(0, 3, 1, 2), ref, what="forward")
    rel_close(wg.grad, w.grad, what="weight grad")
    rel_close(bg.grad, b.grad, what="bias grad")
    for s, x, c in zip(srcs, xs, cins):
        rel_close(s.grad[:, :c].view(B, H, W, c).permute(0, 3, 1, 2), x.grad, what="input grad")
        assert bool((s.grad[:, c:] == 0).all())


@pytest.mark.parametrize("B,H,W,cin,cout,k", [
    (2, 23, 31, 64, 96, 3),        # BasicEncoder layer2.0.conv1 (odd grid: Ho = 12, Wo = 16)
    (1, 24, 18, 96, 128, 3),
    (2, 23, 31, 64, 96, 1),        # the residual block's 1x1 stride-2 downsample
    (1, 16, 40, 32, 32, 3),        # SmallEncoder bottleneck conv2 (32 channels, cout below one 64-row tile)
])
def test_strided_conv_pm_forward_and_gradients(gpu, B, H, W, cin, cout, k):
    """Conv2d(k, stride 2, padding k//2) of the encoders (raft/extractor.py:20-22, 78-80): forward strided in the implicit-GEMM
    kernel, weight gradient strided in the wgrad kernel (no zero-upsampled gradient), data gradient through the zero-upsampled
    gradient — against float64 autograd of F.conv2d(stride=2)."""
    from ptlflow_amd.train import conv_pm
    torch.manual_seed(11)
    x = torch.randn(B, cin, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, dtype=torch.float64) / math.sqrt(cin * k * k)).requires_grad_()
    b = (torch.randn(cout, dtype=torch.float64) * 0.1).requires_grad_()
    ref = F.conv2d(x, w, b, stride=2, padding=k // 2)
    Ho, Wo = ref.shape[2:]
    gout = torch.randn_like(ref)
    ref.backward(gout)
    pm = x.detach().float().permute(0, 2, 3, 1).reshape(B * H * W, cin).cuda().requires_grad_()
    wg = w.detach().float().cuda().requires_grad_()
    bg = b.detach().float().cuda().requires_grad_()
    out = conv_pm([pm], wg, bg, B, H, W, False, [cin], None, 2)
    assert out.shape == (B * Ho * Wo, cout)
    out.backward(gout.float().permute(0, 2, 3, 1).reshape(B * Ho * Wo, cout).cuda())
    rel_close(out.view(B, Ho, Wo, cout).permute(0, 3, 1, 2), ref, what="forward")
    rel_close(wg.grad, w.grad, what="weight grad")
    rel_close(bg.grad, b.grad, what="bias grad")
    rel_close(pm.grad.view(B, H, W, cin).permute(0, 3, 1, 2), x.grad, what="input grad")


@pytest.mark.parametrize("small", [False, True])
def test_update_block_training_step(gpu, small):
    """One update-block call in a training graph: outputs and the gradient of a scalar loss w.r.t. every parameter and
    w.r.t. net / inp / corr, vs float64 autograd through the oracle's functional block."""
    from ptlflow_amd.raft import _param_tree
    from ptlflow_amd.synth import synth_state_dict, update_block_shapes
    from ptlflow_amd.update import PfkUpdateBlock, basic_spec, small_spec
    spec = small_spec() if small else basic_spec()
    holder = _param_tree(update_block_shapes(spec))
    P = synth_state_dict({k: tuple(v.shape) for k, v in holder.state_dict().items()}, seed=19)
    holder.load_state_dict(P)
    ub = PfkUpdateBlock(holder, spec).cuda().train()
    g = torch.Generator().manual_seed(4)
    B, H, W = 2, 12, 18
    net = torch.tanh(torch.randn(B, spec.hidden, H, W, generator=g))
    inp = torch.relu(torch.randn(B, spec.context, H, W, generator=g))
    corr = torch.randn(B, spec.corr_channels, H, W, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 2
    wn, wd, wm = (torch.randn(B, c, H, W, generator=g) for c in (spec.hidden, 2, 576))

    def loss_of(n, m, d):
        l = (n * wn.to(n)).sum() + (d * wd.to(d)).sum()
        return l + ((m * wm.to(m)).sum() if m is not None else 0.0)

    # float64 reference
    Pd = {k: v.double().requires_grad_() for k, v in P.items()}
    rin = [t.double().requires_grad_() for t in (net, inp, corr)]
    step = O.small_update_block if small else O.basic_update_block
    n_ref, m_ref, d_ref = step(Pd, rin[0], rin[1], rin[2], flow.double())
    loss_of(n_ref, m_ref, d_ref).backward()

    gin = [t.cuda().requires_grad_() for t in (net, inp, corr)]
    n, m, d = ub(gin[0], gin[1], gin[2], flow.cuda())
    assert n.requires_grad and d.requires_grad
    rel_close(n, n_ref, what="net"), rel_close(d, d_ref, what="delta")
    if not small:
        rel_close(m, m_ref, what="mask")
    loss_of(n, m, d).backward()
    for name, p in ub.named_parameters():
        assert p.grad is not None, name
        rel_close(p.grad, Pd[name].grad, tol=5e-4, what=name)
    for a, b, nm in zip(gin, rin, ("net", "inp", "corr")):
        rel_close(a.grad, b.grad, tol=5e-4, what="d/d" + nm)


def test_training_graph_falls_back_when_asked(gpu):
    """native_backward=False hands gradient-graph calls to the wrapped module (here: a holder without forward)."""
    from ptlflow_amd.raft import _param_tree
    from ptlflow_amd.synth import update_block_shapes
    from ptlflow_amd.update import PfkUpdateBlock, basic_spec
    spec = basic_spec()
    ub = PfkUpdateBlock(_param_tree(update_block_shapes(spec)), spec).cuda().train()
    ub.native_backward = False
    x = torch.zeros(1, 128, 8, 8, device=gpu, requires_grad=True)
    with pytest.raises(Exception):       # the parameter holder has no forward of its own
        ub(x, x, torch.zeros(1, 324, 8, 8, device=gpu), torch.zeros(1, 2, 8, 8, device=gpu))


def test_wgrad_accumulate_flag_and_flush_node(gpu):
    """`conv_wgrad_unpacked(accumulate=True)` adds into dw / db; the step-scoped accumulation of the recurrent uses of one
    parameter (`ConvPacks.acc` + `_Flush`) hands autograd the same total as the per-use gradients it replaces."""
    import math
    from ptlflow_amd.train import conv_pm, packs_for
    ops = torch.ops.pfk
    g = torch.Generator().manual_seed(4)
    B, H, W, cin, cout = 2, 11, 17, 36, 40
    M = B * H * W
    x = torch.randn(M, cin, generator=g).to(gpu)
    dy = torch.randn(M, cout, generator=g).to(gpu)
    dw1, db1 = torch.empty(cout, cin, 3, 3, device=gpu), torch.empty(cout, device=gpu)
    ops.conv_wgrad_unpacked([x], dy, B, H, W, 3, 3, dw1, db1, [cin], 1)
    dw2, db2 = dw1.clone(), db1.clone()
    ops.conv_wgrad_unpacked([x], dy, B, H, W, 3, 3, dw2, db2, [cin], 1, True)
    assert torch.equal(dw2, dw1 + dw1) and torch.equal(db2, db1 + db1)
    # three recurrent uses of one weight: accumulating entry vs plain autograd
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(gpu)
    b = (torch.randn(cout, generator=g) * 0.1).to(gpu)
    xs = [torch.randn(M, cin, generator=g).to(gpu) for _ in range(3)]

    def grads(accumulate):
        wp, bp = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        cache = {}
        loss = 0.0
        for k, xi in enumerate(xs):
            y = conv_pm([xi], wp, bp, B, H, W, relu=True, packs=packs_for(cache, "c", [wp], accumulate))
            loss = loss + (k + 1) * y.square().mean()
        loss.backward()
        return wp.grad, bp.grad

    (gw0, gb0), (gw1, gb1) = grads(False), grads(True)
    assert float((gw1 - gw0).abs().max()) <= 1e-6 * float(gw0.abs().max()) and float((gb1 - gb0).abs().max()) <= 1e-6 * float(gb0.abs().max())
