#!/usr/bin/env python3
"""Where does a training-step gradient mismatch come from?  Compares, for the raft_small / raft configs of
tests/test_gpu_train_step.py, the gradients at the stage boundaries (d fmap, d cnet output) of the libpfk path against float64
autograd through the oracle, and re-runs the correlation backward alone on the step's own coordinates."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raft_oracle as O
from ptlflow_amd.raft import RAFT
from ptlflow_amd.corr import CorrBlock
from ptlflow_amd.train import sequence_loss, update_block_train_pm, convex_upsample
from ptlflow_amd.train_encoder import encoder_train
import ptlflow_amd
ptlflow_amd.load_native()
gpu = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def run(small, B, H, W, iters):
    model = RAFT(small=small, iters=iters).load_synthetic(21)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = O.smooth_pair(B, H, W, seed=4)
    g = torch.Generator().manual_seed(9)
    gt = torch.randn(B, 2, H, W, generator=g) * 4
    valid = (torch.rand(B, 1, H, W, generator=g) > 0.1).float()
    r = 3 if small else 4
    hdim, cdim = (96, 64) if small else (128, 128)
    # ---------------- float64 oracle, staged
    P = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    xi, pads = O.preprocess(x.double())
    i1, i2 = xi[:, 0], xi[:, 1]
    O._BN_TRAIN = True
    fm = O.encoder(O.sub(P, "fnet"), torch.cat([i1, i2], 0), "instance", small).detach().requires_grad_(True)
    cn = O.encoder(O.sub(P, "cnet"), i1, "none" if small else "batch", small).detach().requires_grad_(True)
    O._BN_TRAIN = False
    pyr = O.correlation_pyramid(fm[:B], fm[B:], 4)
    net, inp = torch.split(cn, [hdim, cdim], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    h, w = i1.shape[-2] // 8, i1.shape[-1] // 8
    c0 = O.coords_grid(B, h, w, torch.float64)
    c1 = c0.clone()
    U = O.sub(P, "update_block")
    step = O.small_update_block if small else O.basic_update_block
    preds, coords_seq, corr_leaves = [], [], []
    for _ in range(iters):
        c1 = c1.detach()
        coords_seq.append(c1.clone())
        corr = O.lookup(pyr, c1, r)
        corr.retain_grad(); corr_leaves.append(corr)
        net, m, d = step(U, net, inp, corr, c1 - c0)
        c1 = c1 + d
        fu = O.upflow8(c1 - c0) if m is None else O.convex_upsample(c1 - c0, m)
        preds.append(O.unpad(fu, pads))
    loss = O.sequence_loss(preds, gt.double(), valid.double())
    loss.backward()
    # ---------------- libpfk, staged the same way (leaves at the encoder outputs)
    model = model.to(gpu).train()
    fm_g = fm.detach().float().to(gpu).requires_grad_(True)
    cn_g = cn.detach().float().to(gpu).requires_grad_(True)
    corr_fn = CorrBlock(fm_g[:B], fm_g[B:], num_levels=4, radius=r)
    net, inp = torch.split(cn_g, [hdim, cdim], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    M = B * h * w
    g0 = c0.float().to(gpu)
    g1 = g0.clone()
    Pg = dict(model.update_block.named_parameters())
    hpm = net.permute(0, 2, 3, 1).reshape(M, hdim)
    ipm = inp.permute(0, 2, 3, 1).reshape(M, cdim)
    preds_g, corr_g = [], []
    cache = {}
    for it in range(iters):
        g1 = g1.detach()
        print(f"  it {it}: coords GPU vs f64 max diff {float((g1.cpu().double() - coords_seq[it]).abs().max()):.2e}")
        cpm = corr_fn.lookup_pm(g1)
        cpm.retain_grad(); corr_g.append(cpm)
        fpm = (g1 - g0).permute(0, 2, 3, 1).reshape(M, 2)
        hpm, mpm, dpm = update_block_train_pm(Pg, model.spec, hpm, ipm, cpm, fpm, B, h, w, cache)
        g1 = g1 + dpm.view(B, h, w, 2).permute(0, 3, 1, 2)
        fl = g1 - g0
        fu = convex_upsample(fl, mpm) if mpm is not None else 8 * F.interpolate(fl, size=(8 * h, 8 * w), mode="bilinear", align_corners=True)
        preds_g.append(model.unpad(fu, pads))
    loss_g = sequence_loss(preds_g, gt.to(gpu), valid.to(gpu))
    loss_g.backward()
    print(f"{'small' if small else 'basic'}: loss {loss_g.item():.6f} vs {loss.item():.6f}")
    for it in range(iters):
        ref = corr_leaves[it].grad.permute(0, 2, 3, 1).reshape(M, -1)
        print(f"  d corr[{it}] rel err {rel(corr_g[it].grad, ref):.2e}")
    print(f"  d cnet-out rel err {rel(cn_g.grad, cn.grad):.2e}")
    print(f"  d fmap1 rel err {rel(fm_g.grad[:B], fm.grad[:B]):.2e}   d fmap2 rel err {rel(fm_g.grad[B:], fm.grad[B:]):.2e}")
    # correlation backward alone: feed the float64 d corr into libpfk's CorrBlock at the float64 coordinates
    f1 = fm.detach().float().to(gpu).requires_grad_(True)
    cb = CorrBlock(f1[:B], f1[B:], num_levels=4, radius=r)
    tot = 0
    for it in range(iters):
        out = cb.lookup_pm(coords_seq[it].float().to(gpu))
        tot = tot + (out * corr_leaves[it].grad.permute(0, 2, 3, 1).reshape(M, -1).float().to(gpu)).sum()
    tot.backward()
    print(f"  corr backward alone (f64 d corr, f64 coords): d fmap1 {rel(f1.grad[:B], fm.grad[:B]):.2e}  d fmap2 {rel(f1.grad[B:], fm.grad[B:]):.2e}")
    # per level: which level's contribution is off?  (oracle per-level grads through its own pyramid)
    fm2 = fm.detach().clone().requires_grad_(True)
    pyr2 = O.correlation_pyramid(fm2[:B], fm2[B:], 4)
    n = 2 * r + 1
    for l in range(4):
        tot2 = 0
        for it in range(iters):
            gl = corr_leaves[it].grad[:, l * n * n:(l + 1) * n * n]
            o = O.lookup(pyr2, coords_seq[it], r)[:, l * n * n:(l + 1) * n * n]
            tot2 = tot2 + (o * gl).sum()
        gref = torch.autograd.grad(tot2, fm2, retain_graph=True)[0]
        f3 = fm.detach().float().to(gpu).requires_grad_(True)
        cb3 = CorrBlock(f3[:B], f3[B:], num_levels=4, radius=r)
        tot3 = 0
        for it in range(iters):
            gfull = torch.zeros_like(corr_leaves[it].grad)
            gfull[:, l * n * n:(l + 1) * n * n] = corr_leaves[it].grad[:, l * n * n:(l + 1) * n * n]
            tot3 = tot3 + (cb3.lookup_pm(coords_seq[it].float().to(gpu)) * gfull.permute(0, 2, 3, 1).reshape(M, -1).float().to(gpu)).sum()
        tot3.backward()
        print(f"    level {l}: d fmap1 {rel(f3.grad[:B], gref[:B]):.2e}  d fmap2 {rel(f3.grad[B:], gref[B:]):.2e}   (scale {float(gref.abs().max()):.2e})")


run(True, 1, 184, 248, 3)
run(False, 2, 368, 496, 3)
