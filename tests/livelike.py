"""TEST INFRASTRUCTURE — a stand-in for a live ptlflow RAFT / GMA model on machines without ptlflow (the GPU box).

`patch.accelerate` dispatches on where a class lives (`ptlflow.models.raft.update.BasicUpdateBlock`, ...) and on its
parameter shapes, and it rebinds the `get_corr_block` global of the model's module.  To exercise exactly that code on the
MI355X — where /root/reference does not exist — this file assembles modules with the reference's module paths, class
names, attribute names and state_dict keys, whose *forward* is the CPU oracle's functional restatement (so, unpatched and on
CPU, the model computes what the reference computes; tests/test_oracle_vs_reference.py pins that equivalence against the
real code).  The caller loop below follows ptlflow/models/raft/raft.py:125-194 and gma/gma.py:141-214: `get_corr_block`
looked up as a module global once per forward, `update_block(net, inp, corr, flow[, attention])` per iteration, `inp` and
the attention map created as fresh tensors per forward.

When the real reference is importable (the build container) the tests use it instead of this file.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn

from oracle import raft_oracle as O
from ptlflow_amd.raft import Encoder, _Attention, _param_tree
from ptlflow_amd.synth import synth_state_dict, update_block_shapes
from ptlflow_amd.update import basic_spec, gma_spec, small_spec


def _module(name: str) -> types.ModuleType:
    mod = sys.modules.get(name)
    if mod is None:
        mod = types.ModuleType(name)
        sys.modules[name] = mod
    return mod


def _place(cls, module: str, name: str):
    cls.__module__, cls.__name__, cls.__qualname__ = module, name, name
    setattr(_module(module), name, cls)
    return cls


class _OracleCorrBlock:
    """What `get_corr_block` returns before patching: the reference CorrBlock's arithmetic (raft/corr.py:12-64)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.radius = radius
        self.corr_pyramid = O.correlation_pyramid(fmap1.float().cpu(), fmap2.float().cpu(), num_levels)
        self.device = fmap1.device

    def __call__(self, coords):
        return O.lookup(self.corr_pyramid, coords.float().cpu(), self.radius).to(self.device)


def _make_update_block(kind: str, module: str, name: str):
    spec = {"basic": basic_spec, "small": small_spec, "gma": gma_spec}[kind]()
    fn = {"basic": O.basic_update_block, "small": O.small_update_block, "gma": O.gma_update_block}[kind]

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            for n, child in _param_tree(update_block_shapes(spec)).named_children():
                self.add_module(n, child)

        def forward(self, net, inp, corr, flow, *extra):
            P = {k: v.detach() for k, v in self.named_parameters()}
            return fn(P, net, inp, corr, flow, *extra)

    return _place(Block, module, name)


def build(kind: str = "raft", iters: int = 32, seed: int = 1234) -> nn.Module:
    """kind in {"raft", "raft_small", "gma"}: a model laid out like ptlflow's, weights seeded, eval mode, on CPU."""
    fam = "gma" if kind == "gma" else "raft"
    small = kind == "raft_small"
    model_mod = _module(f"ptlflow.models.{fam}.{fam}")
    model_mod.get_corr_block = lambda fmap1, fmap2, num_levels=4, radius=4, alternate_corr=False: _OracleCorrBlock(
        fmap1, fmap2, num_levels, radius)
    BasicEncoder = _place(type("BasicEncoder", (Encoder,), {}), f"ptlflow.models.{fam}.extractor", "BasicEncoder")
    SmallEncoder = _place(type("SmallEncoder", (Encoder,), {}), f"ptlflow.models.{fam}.extractor", "SmallEncoder")
    ub_name = {"raft": "BasicUpdateBlock", "raft_small": "SmallUpdateBlock", "gma": "GMAUpdateBlock"}[kind]
    UB = _make_update_block({"raft": "basic", "raft_small": "small", "gma": "gma"}[kind], f"ptlflow.models.{fam}.update", ub_name)

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.corr_levels, self.corr_radius, self.iters = 4, (3 if small else 4), iters
            self.hidden_dim, self.context_dim = (96, 64) if small else (128, 128)
            enc = SmallEncoder if small else BasicEncoder
            self.fnet = enc(128 if small else 256, "instance", small)
            self.cnet = enc(self.hidden_dim + self.context_dim, "none" if small else "batch", small)
            self.update_block = UB()
            if kind == "gma":
                self.att = _Attention(dim=128, heads=1, dim_head=128, max_pos_size=160)

        @torch.no_grad()
        def forward(self, inputs):
            x, pads = O.preprocess(inputs["images"])
            image1, image2 = x[:, 0].contiguous(), x[:, 1].contiguous()
            fmap1, fmap2 = self.fnet([image1, image2])
            # the module GLOBAL, looked up at call time — what `patch.accelerate` rebinds (raft.py:146)
            corr_fn = sys.modules[type(self).__module__].get_corr_block(
                fmap1=fmap1, fmap2=fmap2, radius=self.corr_radius, num_levels=self.corr_levels, alternate_corr=False)
            cnet = self.cnet(image1)
            net, inp = torch.split(cnet, [self.hidden_dim, self.context_dim], dim=1)
            net, inp = torch.tanh(net), torch.relu(inp)                         # fresh tensors every forward
            extra = (self.att(inp),) if kind == "gma" else ()
            B, _, H, W = image1.shape
            coords0 = O.coords_grid(B, H // 8, W // 8).to(x.device)
            coords1 = coords0.clone()
            flow_up = None
            for _ in range(self.iters):
                coords1 = coords1.detach()
                corr = corr_fn(coords1)
                flow = coords1 - coords0
                net, up_mask, delta = self.update_block(net, inp, corr, flow, *extra)
                coords1 = coords1 + delta
                flow_up = O.upflow8(coords1 - coords0) if up_mask is None else O.convex_upsample(coords1 - coords0, up_mask)
                flow_up = O.unpad(flow_up, pads)
            return {"flows": flow_up[:, None], "flow_small": coords1 - coords0}

    Model = _place(Model, f"ptlflow.models.{fam}.{fam}", {"raft": "RAFT", "raft_small": "RAFTSmall", "gma": "GMA"}[kind])
    model = Model()
    own = model.state_dict()
    new = synth_state_dict({k: tuple(v.shape) for k, v in own.items() if v.is_floating_point()}, seed)
    new.update({k: v for k, v in own.items() if not v.is_floating_point()})
    model.load_state_dict(new, strict=True)
    return model.eval()
