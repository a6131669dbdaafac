#!/usr/bin/env python3
"""Encoder weight-gradient errors against float64 for four implementations of the same graph: libpfk (encoder_train), libpfk with
float64 statistics, torch modules on the GPU (MIOpen), torch modules on the CPU in float32."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raft_oracle as O
from ptlflow_amd.raft import Encoder
from ptlflow_amd.synth import synth_state_dict
import ptlflow_amd.train_encoder as TE
import ptlflow_amd
ptlflow_amd.load_native()
gpu = torch.device("cuda:0")


def run(kind, small, B, H, W, noise=False):
    out_dim = 128 if small else 256
    enc = Encoder(out_dim, kind, small)
    sd = synth_state_dict({"fnet." + k: tuple(v.shape) for k, v in enc.state_dict().items()}, 31)
    sd = {k[len("fnet."):]: v for k, v in sd.items()}
    enc.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1 if noise else (O.smooth_pair(B, H, W, seed=8)[:, 0] - 0.5) * 2.0
    ref_mod = Encoder(out_dim, kind, small).double()
    ref_mod.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    ref_mod.train()
    ref = ref_mod(x.double())
    go = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * go).sum().backward()
    gref = {n: p.grad for n, p in ref_mod.named_parameters()}

    def errs(mod):
        return {n: float((p.grad.double().cpu() - gref[n]).abs().max()) / max(float(gref[n].abs().max()), 1e-12)
                for n, p in mod.named_parameters() if p.dim() == 4}

    res = {}
    tags = ("libpfk", "libpfk+bn:miopen", "libpfk+bn:torch", "torch-gpu", "torch-cpu32") if kind == "batch" else ("libpfk", "torch-gpu", "torch-cpu32")
    for tag in tags:
        m = Encoder(out_dim, kind, small)
        m.load_state_dict(sd)
        m.train()
        if tag.startswith("libpfk"):
            TE._DEBUG_BN = tag.split("bn:")[1] if "bn:" in tag else ""
            m = m.to(gpu)
            out = TE.encoder_train(m, x.to(gpu))
            (out * go.float().to(gpu)).sum().backward()
            TE._DEBUG_BN = ""
        elif tag == "torch-gpu":
            m = m.to(gpu)
            (m(x.to(gpu)) * go.float().to(gpu)).sum().backward()
        else:
            (m(x) * go.float()).sum().backward()
        res[tag] = errs(m)
    names = [n for n, p in ref_mod.named_parameters() if p.dim() == 4]    # network order
    print(f"== {kind} small={small} B={B} {H}x{W} {'iid-noise' if noise else 'smooth'} frames: weight-gradient max error / scale vs float64")
    for n in names:
        print(f"  {n:28s} " + "  ".join(f"{t} {res[t][n]:.1e}" for t in res))


run("batch", False, 2, 96, 136, True)
run("instance", False, 2, 96, 136, True)
run("instance", True, 2, 184, 248, True)
