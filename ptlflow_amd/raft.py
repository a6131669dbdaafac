"""Host-side mirror of the reference's RAFT / RAFTSmall callers (ptlflow/models/raft/raft.py:48-234)
for machines where ptlflow itself is not installed (the GPU bench box, CI).

Same constructor arguments that matter on the path, same ``forward(inputs) -> outputs`` dict contract
(``inputs["images"]: [B,2,3,H,W]`` BGR in [0,1]; ``outputs["flows"]: [B,1,2,H,W]``, ``"flow_small"`` in
eval), and the same ``state_dict`` key names / shapes as the reference classes, so
``raft-things-802bbcfd.ckpt``-style checkpoints load unchanged (checked against the live reference in
tests/test_oracle_vs_reference.py).

What runs where:
* encoders (fnet / cnet, raft/extractor.py BasicEncoder) — libpfk kernels through `EncoderEngine` (ptlflow_amd/encoder.py);
  raft_small's bottleneck SmallEncoder and `native_encoders=False` keep the torch modules;
* correlation volume, pyramid, per-iteration lookup, the whole update block, coordinate update and
  convex upsampling — libpfk kernels through torch.ops.pfk, state kept pixel-major across all
  iterations (no NCHW round trips inside the loop).

On a machine that *has* ptlflow, use `ptlflow_amd.patch.accelerate(model)` on the real model instead.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import load_native
from .corr import AlternateCorrBlock, CorrBlock
from .update import UpdateEngine, UpdateSpec, basic_spec, gma_spec, small_spec
from .synth import synth_state_dict, update_block_shapes


# ----------------------------------------------------------------------------- encoders
def _make_norm(kind: str, ch: int) -> nn.Module:
    if kind == "instance":
        return nn.InstanceNorm2d(ch)
    if kind == "batch":
        return nn.BatchNorm2d(ch)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(kind)


class _Block(nn.Module):
    """Residual (3x3,3x3) or bottleneck (1x1,3x3,1x1) unit with the reference's attribute names
    (raft/extractor.py:6-119) so parameters land on identical state_dict keys."""

    def __init__(self, cin: int, cout: int, kind: str, stride: int, bottleneck: bool):
        super().__init__()
        self.bottleneck = bottleneck
        if bottleneck:
            mid = cout // 4
            self.conv1 = nn.Conv2d(cin, mid, 1)
            self.conv2 = nn.Conv2d(mid, mid, 3, padding=1, stride=stride)
            self.conv3 = nn.Conv2d(mid, cout, 1)
            self.norm1, self.norm2, self.norm3 = (_make_norm(kind, c) for c in (mid, mid, cout))
            extra = "norm4"
        else:
            self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, stride=stride)
            self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
            self.norm1, self.norm2 = _make_norm(kind, cout), _make_norm(kind, cout)
            extra = "norm3"
        self.downsample = None
        if stride != 1:
            setattr(self, extra, _make_norm(kind, cout))
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride), getattr(self, extra))

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.bottleneck:
            y = F.relu(self.norm3(self.conv3(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class Encoder(nn.Module):
    """BasicEncoder / SmallEncoder (raft/extractor.py:122-267), stride 8."""

    def __init__(self, output_dim: int, norm_fn: str, small: bool):
        super().__init__()
        dims = (32, 32, 64, 96) if small else (64, 64, 96, 128)
        self.norm_fn = norm_fn
        self.norm1 = _make_norm(norm_fn, dims[0])
        self.conv1 = nn.Conv2d(3, dims[0], 7, stride=2, padding=3)
        cin = dims[0]
        for i, (d, stride) in enumerate(zip(dims[1:], (1, 2, 2)), start=1):
            setattr(self, f"layer{i}", nn.Sequential(_Block(cin, d, norm_fn, stride, small), _Block(d, d, norm_fn, 1, small)))
            cin = d
        self.conv2 = nn.Conv2d(cin, output_dim, 1)

    def forward(self, x):
        is_list = isinstance(x, (tuple, list))      # two frames batched and split again, extractor.py:173-193
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(list(x), dim=0)
        x = F.relu(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        x = self.conv2(x)
        return torch.split(x, [batch_dim, batch_dim], dim=0) if is_list else x


# ----------------------------------------------------------------------------- parameter holder
def _param_tree(shapes: Dict[str, tuple]) -> nn.Module:
    """Nested modules holding nn.Parameters at the dotted names of `shapes`
    (`encoder.convc1.weight`, `mask.0.bias`, ...) — a weights-only stand-in for the reference's
    BasicUpdateBlock whose forward is the kernel chain in `UpdateEngine`."""
    root = nn.Module()
    for name, shape in shapes.items():
        mod = root
        parts = name.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape)))
    return root


# ----------------------------------------------------------------------------- model
class RAFT(nn.Module):
    def __init__(self, corr_levels: int = 4, corr_radius: Optional[int] = None, iters: int = 32, small: bool = False,
                 upsample_every_iter: bool = True, conv_precision: str = "fp32", native_encoders: bool = True,
                 alternate_corr: bool = False, use_graph: bool = False, overlap_mask_head: bool = True,
                 fork_branches: Optional[bool] = None, hoist_context: bool = True, fuse_mask_upsample: Optional[bool] = None):
        super().__init__()
        self.small = small
        # The mask head's second convolution, the softmax over its nine taps and the convex upsampling (raft/update.py:152,
        # raft/raft.py:112-123) as ONE kernel that never writes the [M, 576] mask (`pfk_mask_upsample_f32`); bit-identical
        # to the two separate launches.  None: where it pays — fp32 convolutions from 28 160 pixels up (batch 8: 196 us against
        # 173 + 41 for the pair, 254 MB less HBM traffic per iteration; at batch 1 the two forms take the same 34 us);
        # True / False: always (where the kernel applies) / never.
        self.fuse_mask_upsample = fuse_mask_upsample
        # True (default): the context features' part of the GRU convolutions — loop-invariant, `inp` is the same tensor in every
        # iteration (raft.py:158-160, update.py:60-71) — is computed once per forward instead of once per iteration
        # (UpdateEngine's docstring); False: the single-chain launches (A/B, tests)
        self.hoist_context = hoist_context
        # True: the mask head's second convolution and the convex upsampling of iteration i — off the recurrent critical path:
        # nothing of iteration i+1 reads their results — run on a second HIP stream next to iteration i+1's lookup / motion
        # encoder / GRU, filling the idle CUs of those launches' tails (same kernels, same operands: bit-identical output).
        # Eager loop only; the hipGraph path keeps one stream.
        self.overlap_mask_head = overlap_mask_head
        # None: fork the independent branches of an iteration (flow branch of the motion encoder; mask conv2 + upsampling) onto
        # side streams only while the loop is being captured into a hipGraph; True / False: always / never (`_iterate`)
        self.fork_branches = fork_branches
        self._side_streams: Dict[tuple, torch.cuda.Stream] = {}
        # True: the 32-iteration loop (lookup, update block, upsampling: ~20 launches per iteration) is captured once per
        # input shape into a hipGraph (torch.cuda.CUDAGraph) and replayed — it runs entirely on buffers with fixed addresses —
        # with the independent branches of an iteration forked onto concurrent graph branches (`_iterate_forked`); models with the
        # materialised volume only (no GMA aggregate, no alternate_corr).  The first forward of a shape runs eagerly and records,
        # later ones replay (bit-identical).  Opt-in: measured on MI355X / ROCm 7 at batch 1 of 436x1024 (round 3) it buys
        # nothing — 19.04 ms replayed vs 19.06 ms eager: the loop is execution-bound, kernel-to-kernel dependency gaps are the same
        # inside a graph, and forked graph branches did not run concurrently (19.19 ms); eager forks on real streams gained
        # 0.3 ms only once every block was 48 KB (three per CU), which itself costs 0.5 ms.
        self.use_graph = use_graph
        self.group_launches = None   # None: grouped launches of the motion encoder below 28 160 pixels (`_iterate`); True / False: forced
        self.max_graphs = 4          # recorded shapes kept (oldest dropped first): each holds its loop buffers and pyramid
        self._graphs: Dict[tuple, dict] = {}
        # True: never materialise the N x N volume, compute the lookup windows on demand (raft.py `alternate_corr`,
        # raft/corr.py:67-101) — the memory / time trade for high resolutions
        self.alternate_corr = alternate_corr
        # True: fnet / cnet run on libpfk kernels (ptlflow_amd/encoder.py: BasicEncoder, and raft_small's bottleneck
        # SmallEncoder); False: the torch modules
        self.native_encoders = native_encoders
        self._enc = None
        self._enc_versions = None
        # "fp32" (default: fp32 matrix cores, the parity path) | "bf16x6" | "bf16x3" | "bf16": split-bf16 operands
        # for the update block's convolutions (include/pfk.h, pfk_conv2d_bf16s)
        self.conv_precision = conv_precision
        self.corr_levels = corr_levels
        self.corr_radius = corr_radius if corr_radius is not None else (3 if small else 4)
        self.iters = iters
        # The reference computes mask head + convex upsampling on every iteration even in eval, where only
        # the last one is observable (raft.py:180-187).  True = do the same work; False = skip the dead work.
        self.upsample_every_iter = upsample_every_iter
        if small:
            self.hidden_dim, self.context_dim = 96, 64
            self.fnet = Encoder(128, "instance", True)
            self.cnet = Encoder(self.hidden_dim + self.context_dim, "none", True)
            self.spec: UpdateSpec = small_spec(corr_levels, self.corr_radius)
        else:
            self.hidden_dim, self.context_dim = 128, 128
            self.fnet = Encoder(256, "instance", False)
            self.cnet = Encoder(self.hidden_dim + self.context_dim, "batch", False)
            self.spec = self._basic_spec(corr_levels, self.corr_radius)
        self.update_block = _param_tree(update_block_shapes(self.spec))
        self._engine: Optional[UpdateEngine] = None
        self._versions = None

    def _basic_spec(self, corr_levels: int, corr_radius: int) -> UpdateSpec:
        return basic_spec(corr_levels, corr_radius)

    def _after_context(self, eng: UpdateEngine, inp: torch.Tensor) -> None:
        """Hook between the context network and the loop (GMA computes its attention map here)."""

    # -- weights ---------------------------------------------------------------------------------
    def load_synthetic(self, seed: int = 1234) -> "RAFT":
        own = self.state_dict()
        shapes = {k: tuple(v.shape) for k, v in own.items() if v.is_floating_point()}
        new = synth_state_dict(shapes, seed)
        new.update({k: v for k, v in own.items() if not v.is_floating_point()})   # index buffers / counters stay
        self.load_state_dict(new, strict=True)
        return self

    def engine(self, device) -> UpdateEngine:
        params = dict(self.update_block.named_parameters())
        v = tuple((p.data_ptr(), p._version) for p in params.values())
        if self._engine is None or self._engine.device != device:
            self._engine = UpdateEngine(params, self.spec, device, self.conv_precision, self.hoist_context)
        elif v != self._versions:
            # re-packing allocates new weight tensors: every captured graph still points at the old (freed) ones
            self._engine.pack(params)
            self._graphs.clear()
        self._versions = v
        return self._engine

    def encoders(self, device):
        """(fnet, cnet) callables: EncoderEngine pair on libpfk kernels, or the torch modules."""
        if not self.native_encoders:
            return self.fnet, self.cnet
        from .encoder import EncoderEngine
        v = tuple((p.data_ptr(), p._version) for m in (self.fnet, self.cnet) for p in list(m.parameters()) + list(m.buffers()))
        if self._enc is None or self._enc[0].device != device or v != self._enc_versions:
            self._enc = (EncoderEngine(self.fnet.state_dict(), "instance", device, self.conv_precision, self.small),
                         EncoderEngine(self.cnet.state_dict(), "none" if self.small else "batch", device, self.conv_precision, self.small))
            self._enc_versions = v
        return self._enc

    # -- pre/post-processing (raft.py:127-135 -> base_model.py:207-247; utils/external/raft.py:57-84)
    @staticmethod
    def _pads(ht: int, wd: int, stride: int = 8):
        ph = (((ht // stride) + 1) * stride - ht) % stride
        pw = (((wd // stride) + 1) * stride - wd) % stride
        return (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)

    def preprocess(self, images: torch.Tensor):
        x = (images + (-0.5)) * 2.0
        x = torch.flip(x, [-3])
        pads = self._pads(x.shape[-2], x.shape[-1])
        shp = x.shape
        x = F.pad(x.reshape(-1, *shp[-3:]), pads, mode="replicate")
        return x.reshape(*shp[:-2], *x.shape[-2:]).contiguous(), pads

    @staticmethod
    def unpad(x: torch.Tensor, pads):
        ht, wd = x.shape[-2:]
        return x[..., pads[2]: ht - pads[3], pads[0]: wd - pads[1]]

    # -- forward -----------------------------------------------------------------------------------
    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self.training:
            return self._forward_train(inputs)
        with torch.no_grad():
            return self._forward_eval(inputs)

    def _forward_train(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """RAFT.forward in training mode (raft.py:125-193): same loop, every iteration's upsampled flow kept in
        ``flow_preds`` for the sequence loss (raft.py:20-45; `ptlflow_amd.train.sequence_loss`).  The correlation volume,
        pyramid and lookups (forward and backward), the whole update block (forward, dgrad, wgrad) and the convex upsampling
        (forward and backward) and both encoders (ptlflow_amd/train_encoder.py; BatchNorm of `cnet` with batch statistics, as
        `model.train()` implies) run on libpfk through autograd nodes (ptlflow_amd/corr.py, ptlflow_amd/train.py);
        `native_encoders=False` keeps the encoders on the torch modules."""
        from .train import convex_upsample, update_block_train_pm
        from .train_encoder import encoder_train
        load_native()
        images = inputs["images"]
        if not images.is_cuda:
            raise RuntimeError("ptlflow_amd.RAFT needs GPU inputs (no CPU fallback)")
        x, pads = self.preprocess(images.float())
        image1, image2 = x[:, 0].contiguous(), x[:, 1].contiguous()
        B = image1.shape[0]
        if self.native_encoders:      # forward, data and weight gradients of both encoders on libpfk (ptlflow_amd/train_encoder.py)
            fm = encoder_train(self.fnet, torch.cat([image1, image2], 0))
            fmap1, fmap2 = fm[:B], fm[B:]
            cnet = encoder_train(self.cnet, image1)
        else:                         # the torch modules (MIOpen convolutions / norms under torch.autograd)
            fmap1, fmap2 = self.fnet([image1, image2])
            cnet = self.cnet(image1)
        corr_fn = CorrBlock(fmap1.float(), fmap2.float(), num_levels=self.corr_levels, radius=self.corr_radius)
        net, inp = torch.split(cnet, [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        h, w = image1.shape[-2] // 8, image1.shape[-1] // 8
        M = B * h * w
        ys, xs = torch.meshgrid(torch.arange(h, device=x.device, dtype=torch.float32),
                                torch.arange(w, device=x.device, dtype=torch.float32), indexing="ij")
        coords0 = torch.stack([xs, ys], 0)[None].repeat(B, 1, 1, 1).contiguous()
        coords1 = coords0.clone()
        prev = inputs.get("prev_preds")
        if prev is not None and prev.get("flow_small") is not None:      # warm start applies in training mode too (raft.py:162-167)
            fwd = torch.empty_like(coords1)
            torch.ops.pfk.forward_interpolate(prev["flow_small"].detach().to(device=x.device, dtype=torch.float32).contiguous(), fwd)
            coords1 = coords1 + fwd
        P = dict(self.update_block.named_parameters())
        cache: dict = {}
        # GMA (gma.py:176-177): the attention map of the context features, torch ops under autograd (`_Attention.forward`)
        attention = self.att(inp) if self.spec.aggregate else None
        hpm = net.permute(0, 2, 3, 1).reshape(M, self.hidden_dim)
        ipm = inp.permute(0, 2, 3, 1).reshape(M, self.context_dim)
        preds = []
        for _ in range(self.iters):
            coords1 = coords1.detach()
            corr_pm = corr_fn.lookup_pm(coords1)
            fpm = (coords1 - coords0).permute(0, 2, 3, 1).reshape(M, 2)
            hpm, mask_pm, delta_pm = update_block_train_pm(P, self.spec, hpm, ipm, corr_pm, fpm, B, h, w, cache,
                                                           accumulate_wgrad=True,      # `cache` lives for this step only
                                                           attention=attention)
            coords1 = coords1 + delta_pm.view(B, h, w, 2).permute(0, 3, 1, 2)
            flow = coords1 - coords0
            if mask_pm is not None:
                flow_up = convex_upsample(flow, mask_pm)
            else:
                flow_up = 8 * F.interpolate(flow, size=(8 * h, 8 * w), mode="bilinear", align_corners=True)
            preds.append(self.unpad(flow_up, pads))
        return {"flows": preds[-1][:, None], "flow_preds": preds}

    def _forward_eval(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        load_native()
        ops = torch.ops.pfk
        images = inputs["images"]
        if not images.is_cuda:
            raise RuntimeError("ptlflow_amd.RAFT needs GPU inputs (no CPU fallback)")
        x, pads = self.preprocess(images.float())
        image1, image2 = x[:, 0], x[:, 1]
        B = image1.shape[0]

        fnet, cnet_fn = self.encoders(x.device)
        # the context network only needs frame 1 and shares nothing with the feature network: in the small-batch regime (where
        # the encoders' launches do not fill the chip) it runs on a side stream next to fnet + the correlation volume
        enc_side = None
        if self.native_encoders and B * image1.shape[-2] * image1.shape[-1] <= 2 * 440 * 1024:
            main = torch.cuda.current_stream(x.device)
            enc_side = self._stream(x.device, "enc")
            enc_side.wait_event(main.record_event())
            with torch.cuda.stream(enc_side):
                cnet = cnet_fn(image1)
                cnet_done = enc_side.record_event()
        fm = fnet(torch.cat([image1, image2], 0))
        h, w = image1.shape[-2] // 8, image1.shape[-1] // 8
        eng = self.engine(x.device)
        eng.bind(B, h, w)
        # (per-launch HIP-event instrumentation — bench.py's roofline leg — needs real launches)
        graphable = self.use_graph and not self.spec.aggregate and not self.alternate_corr and eng.profile is None
        # everything the recorded launch sequence depends on besides the buffers' addresses
        gkey = (B, h, w, x.device, self.iters, self.upsample_every_iter, self.corr_levels, self.corr_radius)
        st = self._graphs.get(gkey) if graphable else None
        if st is not None and (st["engine"] is not eng or st["hx_ptr"] != eng.hx.data_ptr()):
            st = None      # a new engine, or its buffers were re-bound for another shape in between: the recorded addresses are stale
        if st is None:
            if self.alternate_corr:
                corr_fn = AlternateCorrBlock(fm[:B], fm[B:], num_levels=self.corr_levels, radius=self.corr_radius,
                                             map_dtype=torch.bfloat16 if self.conv_precision == "bf16" else None)
            else:   # conv_precision "bf16" = BASELINE config 3's precision: bf16 operands everywhere autocast would put them
                vt = torch.bfloat16 if self.conv_precision == "bf16" else torch.float32
                corr_fn = CorrBlock(fm[:B], fm[B:], num_levels=self.corr_levels, radius=self.corr_radius, volume_dtype=vt)
            ys, xs = torch.meshgrid(torch.arange(h, device=x.device, dtype=torch.float32),
                                    torch.arange(w, device=x.device, dtype=torch.float32), indexing="ij")
            coords0 = torch.stack([xs, ys], 0)[None].repeat(B, 1, 1, 1).contiguous()
            coords1 = coords0.clone()
            flow_up = torch.empty(B, 2, 8 * h, 8 * w, device=x.device, dtype=torch.float32)
        else:
            corr_fn, coords0, coords1, flow_up = st["corr"].update(fm[:B], fm[B:]), st["coords0"], st["coords1"], st["flow_up"]
            coords1.copy_(coords0)
        if enc_side is None:
            cnet = cnet_fn(image1)
        else:
            torch.cuda.current_stream(x.device).wait_event(cnet_done)
            cnet.record_stream(torch.cuda.current_stream(x.device))     # allocated on the side stream, consumed here
        net, inp = torch.split(cnet, [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)

        prev = inputs.get("prev_preds")
        if prev is not None and prev.get("flow_small") is not None:      # warm start, raft.py:162-167, on the device
            fwd = torch.empty_like(coords1)
            ops.forward_interpolate(prev["flow_small"].to(device=x.device, dtype=torch.float32).contiguous(), fwd)
            coords1.add_(fwd)

        eng.watch_faults()          # a timed-out stream-K fix-up of an earlier forward raises here (no host sync)
        eng.load_state(net, inp)
        self._after_context(eng, inp)
        ops.flow_from_coords(coords0, coords1, eng.flow_view)
        eng.flow_changed()
        if st is not None:
            st["graph"].replay()
        else:
            flow_up = self._iterate(corr_fn, eng, coords0, coords1, flow_up)
            if graphable:
                # the eager pass above produced this forward's result and warmed every kernel up; record the same loop for
                # the next forwards of this shape (capture does not execute, state is left as the eager pass wrote it)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._iterate(corr_fn, eng, coords0, coords1, flow_up)
                while len(self._graphs) >= self.max_graphs:
                    self._graphs.pop(next(iter(self._graphs)))
                self._graphs[gkey] = {"graph": graph, "corr": corr_fn, "coords0": coords0, "coords1": coords1,
                                                     "flow_up": flow_up, "engine": eng, "hx_ptr": eng.hx.data_ptr()}
        out_up = self.unpad(flow_up, pads)
        flow_small = coords1 - coords0
        if graphable:      # the buffers behind these results are overwritten by the next forward
            out_up = out_up.clone()
        return {"flows": out_up[:, None], "flow_small": flow_small}

    def _stream(self, dev: torch.device, which: str) -> "torch.cuda.Stream":
        key = (dev, which)
        st = self._side_streams.get(key)
        if st is None:
            st = self._side_streams[key] = torch.cuda.Stream(device=dev)
        return st

    def _iterate(self, corr_fn, eng: UpdateEngine, coords0, coords1, flow_up):
        """The recurrent loop of raft.py:169-187 on fixed buffers: lookup -> update block -> coordinate update -> upsampling.

        Three schedules of the SAME launches on the same operands (bit-identical results, tests/test_gpu_model.py):
        * serial — everything on the current stream;
        * mask branch on a side stream (eager, >= 28160 pixels): mask conv2 + convex upsampling of iteration i next to iteration i+1;
        * forked (`fork_branches`, meant for the captured hipGraph where a fork / join is a graph edge, not a host call): the
          independent branches of one iteration run concurrently — the motion encoder's flow branch (convf1 -> convf2, reads
          only the flow slice) next to lookup -> convc1 -> convc2; mask conv2 + upsampling next to the coordinate update and
          the next iteration — so the 110..440-tile launches of the batch-1 regime fill each other's idle CUs
          (update.py:104-112: `cor` and `flo` only meet in `conv`; :152: the mask head reads `net` only)."""
        ops = torch.ops.pfk
        has_mask = self.spec.has_mask
        h, w = coords0.shape[-2:]
        dev = coords0.device
        capturing = torch.cuda.is_current_stream_capturing()
        pixels = coords0.shape[0] * h * w
        fork = self.fork_branches if self.fork_branches is not None else capturing
        if fork and not self.spec.aggregate and not self.alternate_corr:
            return self._iterate_forked(corr_fn, eng, coords0, coords1, flow_up)
        side = None
        # (measured, one MI355X: +0.9 % at batch 8 of 436x1024; -1.4 % at batch 1, where the loop's launches are short and the extra
        #  events cost more than the filled tails give back: only for >= 4 x 7040 pixels)
        fuse = has_mask and eng.can_fuse_mask_upsample and \
            (self.fuse_mask_upsample if self.fuse_mask_upsample is not None else pixels >= 28160)
        # (K8b with the fused K13b: 55 us of mostly matrix / VALU work per iteration — on the main stream it measured 26.44 ms per
        #  batch-8 forward against 26.76 on the side stream, where it competes with the HBM-bound bf16 launches beside it)
        if has_mask and self.overlap_mask_head and pixels >= 28160 and not capturing and not (fuse and eng.b16):
            side = self._stream(dev, "mask")
        main = torch.cuda.current_stream(dev)
        side_done = None
        # Small batches (below the side-stream threshold; fp32 tile kernel): convc1 | convf2 | the PREVIOUS iteration's mask conv2 as one
        # grouped launch (`UpdateEngine.motion_grouped`), the previous iteration's upsampling right behind it — before this
        # iteration's coordinate update overwrites the flow slice it reads.  Same launches' tiles, same bits.
        grouped = side is None and not fuse and eng.can_group and not self.alternate_corr and \
            (self.group_launches if self.group_launches is not None else pixels < 28160)
        owed = False        # the previous iteration's mask conv2 + upsampling have not run yet
        for it in range(self.iters):
            last = it == self.iters - 1
            if eng.profile is not None:     # bench.py's instrumented forward: HIP events around the lookup too (the HBM-bound kernel)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                corr_pm = corr_fn.lookup_pm(coords1, out=eng.lookup_out)
                e1.record()
                eng.profile.setdefault("lookup", []).append((e0, e1))
                n = 2 * self.corr_radius + 1
                eng.flops["lookup"] = 0.0
                # N L [(2r+2)^2 + (2r+1)^2] 4 + 8 N bytes per pair and lookup
                # (element sizes as stored: bf16 maps / bf16 rows on the K8b path)
                eng.bytes["lookup"] = float(pixels * (self.corr_levels * ((n + 1) ** 2 * corr_fn.volume_dtype.itemsize
                                                                        + n * n * corr_pm.element_size()) + 8))
            else:
                corr_pm = corr_fn.lookup_pm(coords1, out=eng.lookup_out)      # K8b engines take the lookup in bf16, written directly
            do_up = last or self.upsample_every_iter
            if grouped:
                eng.motion_and_gru(corr_pm, grouped=True, prev_mask=owed)
                if owed:         # `eng.mask` = mask(it - 1) now; the flow slice still holds flow(it - 1)
                    ops.convex_upsample_pm(eng.flow_view, eng.mask, flow_up)
                    owed = False
                eng.heads(coords0, coords1, None, want_mask=do_up, mask_conv2=last)
                if do_up:
                    if not has_mask:
                        ops.upflow8(coords0, coords1, flow_up)
                    elif last:
                        ops.convex_upsample_pm(eng.flow_view, eng.mask, flow_up)
                    else:
                        owed = True
                continue
            if side is None or not do_up:
                if fuse and do_up:
                    eng.motion_and_gru(corr_pm)
                    eng.heads_conv1(True)
                    eng.flow_delta(coords0, coords1)
                    eng.mask_upsample(flow_up)       # mask conv2 + softmax + convex upsampling, no [M, 576] mask in memory
                    continue
                eng.step(corr_pm, coords0, coords1, want_mask=do_up)
                if do_up:
                    if has_mask:   # flow = coords1 - coords0 is already in the engine's hx slice (written by flow_delta)
                        ops.convex_upsample_pm(eng.flow_view, eng.mask, flow_up)
                    else:          # raft_small: upflow8 (raft/utils.py:94-96)
                        ops.upflow8(coords0, coords1, flow_up)
                continue
            eng.motion_and_gru(corr_pm)
            if side_done is not None:      # the previous iteration's mask head / upsampling still read `fm` and the flow slice
                main.wait_event(side_done)
            eng.heads_conv1(True)                    # fh | mask conv1 (`fm`): the mask half is consumed on the side stream
            eng.flow_delta(coords0, coords1)         # flow head conv2 + coordinate update
            forked = main.record_event()
            side.wait_event(forked)
            with torch.cuda.stream(side):
                if fuse:
                    eng.mask_upsample(flow_up)
                else:
                    eng.mask_head(side_stream=True)
                    ops.convex_upsample_pm(eng.flow_view, eng.mask, flow_up)
                side_done = side.record_event()
        if side_done is not None:
            main.wait_event(side_done)
        return flow_up

    def _iterate_forked(self, corr_fn, eng: UpdateEngine, coords0, coords1, flow_up):
        ops = torch.ops.pfk
        has_mask = self.spec.has_mask
        dev = coords0.device
        main = torch.cuda.current_stream(dev)
        s_flow, s_mask = self._stream(dev, "flow"), self._stream(dev, "mask")
        mask_done = None
        for it in range(self.iters):
            last = it == self.iters - 1
            do_up = last or self.upsample_every_iter
            # fork: the flow branch needs the flow slice of hx (written by the previous coordinate update, on main)
            s_flow.wait_event(main.record_event())
            with torch.cuda.stream(s_flow):
                eng.motion_flow(branch=True)
                flow_done = s_flow.record_event()
            corr_pm = corr_fn.lookup_pm(coords1, out=eng.lookup_out)
            eng.motion_corr(corr_pm)
            main.wait_event(flow_done)                                # join: `conv` reads both halves of corflo
            eng.motion_join()
            eng.gru()
            if mask_done is not None:      # the previous mask branch still reads the mask half of `fm` and the flow slice
                main.wait_event(mask_done)
            eng.heads_conv1(want_mask=do_up)
            if has_mask and do_up:
                s_mask.wait_event(main.record_event())
                with torch.cuda.stream(s_mask):
                    eng.mask_head(side_stream=True)                   # next to the coordinate update
                eng.flow_delta(coords0, coords1)
                s_mask.wait_event(main.record_event())                # the upsampling reads the NEW flow slice
                with torch.cuda.stream(s_mask):
                    ops.convex_upsample_pm(eng.flow_view, eng.mask, flow_up)
                    mask_done = s_mask.record_event()
            else:
                eng.flow_delta(coords0, coords1)
                if do_up and not has_mask:
                    ops.upflow8(coords0, coords1, flow_up)
        if mask_done is not None:
            main.wait_event(mask_done)
        return flow_up


class RAFTSmall(RAFT):
    def __init__(self, **kw):
        kw.setdefault("small", True)
        super().__init__(**kw)


class _RelPosEmb(nn.Module):
    """Parameter holder for GMA's relative position embedding (gma_utils.py:6-30).  The `gma` model registers it
    but never uses it in its default content-only attention; it exists here so checkpoints load with strict=True."""

    def __init__(self, max_pos_size: int, dim_head: int):
        super().__init__()
        self.rel_height = nn.Embedding(2 * max_pos_size - 1, dim_head)
        self.rel_width = nn.Embedding(2 * max_pos_size - 1, dim_head)
        deltas = torch.arange(max_pos_size).view(1, -1) - torch.arange(max_pos_size).view(-1, 1)
        self.register_buffer("rel_ind", deltas + max_pos_size - 1)


class _Attention(nn.Module):
    """Parameter holder + torch fall-back for GMA's attention, content-only mode (gma_utils.py:33-78): one [B,1,N,N] softmax
    map per forward.  `GMA._attention` computes it on libpfk (to_qk on the MFMA conv kernel, the N x N similarity on K1, the
    row softmax kernel); this module's own `forward` (torch ops) only serves several heads or a width that is not a multiple
    of 32."""

    def __init__(self, dim: int = 128, heads: int = 1, dim_head: int = 128, max_pos_size: int = 160):
        super().__init__()
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_qk = nn.Conv2d(dim, heads * dim_head * 2, 1, bias=False)
        self.pos_emb = _RelPosEmb(max_pos_size, dim_head)

    def forward(self, fmap: torch.Tensor) -> torch.Tensor:
        B, _, h, w = fmap.shape
        q, k = self.to_qk(fmap).chunk(2, dim=1)
        d = q.shape[1] // self.heads
        q = q.reshape(B, self.heads, d, h * w).transpose(2, 3) * self.scale
        k = k.reshape(B, self.heads, d, h * w)
        return torch.softmax(torch.matmul(q, k), dim=-1)


class GMA(RAFT):
    """Mirror of ptlflow/models/gma/gma.py:51-214 (`gma`, one head, content-only attention): RAFT's loop with the
    update block of gma/update.py:127-160.  The per-iteration `attn @ v` (12.7 GFLOP, reads the 198 MB map) runs on
    the same MFMA GEMM kernel as the convolutions, with the residual `fmap + gamma * out` fused in its epilogue."""

    def __init__(self, corr_levels: int = 4, corr_radius: int = 4, iters: int = 32, upsample_every_iter: bool = True,
                 conv_precision: str = "fp32"):
        super().__init__(corr_levels=corr_levels, corr_radius=corr_radius, iters=iters, small=False,
                         upsample_every_iter=upsample_every_iter, conv_precision=conv_precision)
        self.att = _Attention(dim=self.context_dim, heads=1, dim_head=self.context_dim, max_pos_size=160)

    def _basic_spec(self, corr_levels: int, corr_radius: int) -> UpdateSpec:
        return gma_spec(corr_levels, corr_radius)

    def _after_context(self, eng: UpdateEngine, inp: torch.Tensor) -> None:
        eng.set_attention(self._attention(inp))

    def _attention(self, inp: torch.Tensor) -> torch.Tensor:
        """Attention.forward, content-only, one head (gma_utils.py:50-78) on libpfk: `to_qk` as two 1x1 convolutions on the
        MFMA conv kernel (q and k land in their own pixel-major matrices), the N x N similarity as K1 (`pfk_corr_volume_f32`
        with scale = dim_head^-0.5: the same all-pairs inner product as the correlation volume), and an in-place row softmax."""
        from .packing import pack_conv_weight
        ops = torch.ops.pfk
        B, C, h, w = inp.shape
        N = h * w
        att = self.att
        if att.heads != 1 or C % 32:
            return att(inp)
        wv = att.to_qk.weight
        key = (wv.data_ptr(), wv._version, wv.device)
        if getattr(self, "_qk_key", None) != key:
            wq, wk = wv.detach().float().chunk(2, dim=0)
            self._qk_w = (pack_conv_weight(wq.contiguous(), [(0, C, C)]), pack_conv_weight(wk.contiguous(), [(0, C, C)]))
            self._qk_key = key
        x = inp.float().permute(0, 2, 3, 1)
        x = x.reshape(B * N, C) if x.is_contiguous() else x.contiguous().view(B * N, C)
        d = wv.shape[0] // 2
        q = torch.empty(B * N, d, device=inp.device, dtype=torch.float32)
        k = torch.empty(B * N, d, device=inp.device, dtype=torch.float32)
        for wt, dst in zip(self._qk_w, (q, k)):
            ops.conv2d([x], B, h, w, 1, 1, wt, None, d, 0, False, 1.0, dst, None, None, None, None)
        attn = torch.empty(B, N, N, device=inp.device, dtype=torch.float32)
        ops.corr_volume(q.view(B, N, d), k.view(B, N, d), float(att.scale), attn)
        ops.softmax_rows(attn)
        return attn.view(B, 1, N, N)
