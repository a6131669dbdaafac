#!/usr/bin/env python3
"""Update-block training step (forward + backward through `iters` recurrent calls) on the libpfk training path, GPU box.
    python scripts/train_bench.py [--batch 8] [--height 368] [--width 496] [--iters 12] [--torch]
--torch runs the same composition with torch.nn.functional.conv2d (MIOpen) instead, for a same-machine comparison."""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd.train as T  # noqa: E402
from ptlflow_amd.raft import _param_tree  # noqa: E402
from ptlflow_amd.synth import synth_state_dict, update_block_shapes  # noqa: E402
from ptlflow_amd.update import basic_spec  # noqa: E402


def torch_conv_pm(srcs, weight, bias, B, H, W, relu=False, real=None, packs=None):
    real = [s.shape[1] for s in srcs] if real is None else real
    x = torch.cat([s[:, :n] for s, n in zip(srcs, real)], 1)
    x = x.view(B, H, W, -1).permute(0, 3, 1, 2)
    y = F.conv2d(x, weight, bias, padding=(weight.shape[2] // 2, weight.shape[3] // 2))
    y = F.relu(y) if relu else y
    return y.permute(0, 2, 3, 1).reshape(B * H * W, -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=368)
    ap.add_argument("--width", type=int, default=496)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--torch", action="store_true")
    args = ap.parse_args()
    if args.torch:
        T.conv_pm = torch_conv_pm
    dev = torch.device("cuda")
    spec = basic_spec()
    holder = _param_tree(update_block_shapes(spec)).to(dev)
    holder.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in holder.state_dict().items()}, seed=3))
    P = dict(holder.named_parameters())
    B, H, W = args.batch, args.height // 8, args.width // 8
    g = torch.Generator(device="cpu").manual_seed(0)
    net0 = torch.tanh(torch.randn(B, 128, H, W, generator=g)).to(dev).requires_grad_()
    inp = torch.relu(torch.randn(B, 128, H, W, generator=g)).to(dev).requires_grad_()
    corrs = [torch.randn(B, 324, H, W, generator=g).to(dev).requires_grad_() for _ in range(2)]
    flow = torch.zeros(B, 2, H, W, device=dev)

    cache = {}

    def step():
        net, loss = net0, 0.0
        f = flow
        for it in range(args.iters):
            net, mask, delta = T.update_block_train(P, spec, net, inp, corrs[it % 2], f, cache)
            f = (f + delta).detach()
            loss = loss + delta.abs().mean() + 1e-3 * mask.abs().mean()
        loss.backward()
        for p in P.values():
            p.grad = None

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    flops = 3 * 6.24e6 * B * H * W * args.iters          # 6.24 MFLOP per pixel per iteration forward; x3 with dgrad + wgrad
    print(f"{'torch/MIOpen' if args.torch else 'libpfk'} update-block training step: batch {B}, {H}x{W} grid, {args.iters} iterations: "
          f"{ms:.1f} ms  ({flops / ms / 1e9:.1f} TFLOP/s of convolution work)")


if __name__ == "__main__":
    main()
