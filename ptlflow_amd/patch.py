"""Drop-in shim for a *live* ptlflow model (a machine that has ptlflow + lightning installed).

    import ptlflow, ptlflow_amd.patch
    model = ptlflow.get_model("raft", ckpt_path="things").eval().cuda()
    ptlflow_amd.patch.accelerate(model)          # <- the only new line
    out = model({"images": images})              # validate.py / infer.py / model_benchmark.py unchanged

What gets replaced:
* seam B1 — every model family binds ``get_corr_block`` by name into its own model module at import
  (`from .corr import get_corr_block`, raft.py:10, gma.py:11, sea_raft.py:10, ccmr.py:10, ms_raft_plus.py:12), so the
  patch target is the *model module's* global, not `corr.py`;
* seam B3 — ``model.update_block`` is wrapped by `PfkUpdateBlock`, which keeps the original sub-modules
  (state_dict keys, checkpoints and optimizers are untouched) and only overrides ``forward``;
* seam B4 — ``model.fnet`` / ``model.cnet`` (`BasicEncoder`) are wrapped by `PfkEncoder` the same way;
* seam B5 — ``model.upsample_flow`` (raft.py:112-123, a method the loop calls every iteration) is shadowed on the instance by
  the convex-upsampling kernel after a behaviour probe (`_UpsampleSeam`).

Dispatch is by *implementation identity*, never by class name alone: about two dozen ptlflow families call their block
``BasicUpdateBlock`` with different layers and ``forward`` signatures (sea_raft/update.py:39-54 is a ConvNeXt stack that
returns one tensor; ccmr/update.py:152 takes six arguments).  A block is wrapped only when
  (1) its class comes from a module whose source was checked to be the RAFT/GMA implementation (`_UPDATE_BLOCKS`,
      `_ENCODERS` below — extendable with `register_update_block` / `register_encoder`), AND
  (2) its ``state_dict`` names and shapes are exactly those of the `UpdateSpec` / `BasicEncoder` the kernels implement.
Everything else is left untouched and keeps running the reference's own code; the B1 hook (which only depends on the
`get_corr_block` contract the five families share) is still installed for it.

Nothing in ptlflow is edited or copied; `restore(model)` undoes the patch.
"""
from __future__ import annotations

import contextvars
import importlib
import sys
from typing import Callable, Dict, Optional, Tuple

import torch

from . import load_native
from .corr import get_corr_block as _pfk_get_corr_block
from .update import PfkUpdateBlock, UpdateSpec, basic_spec, ccmr_spec, gma_spec, ms_raft_plus_spec, small_spec

_ORIG = "_pfk_original_get_corr_block"

# (module of the class, class name) -> spec factory(corr_channels); raft/update.py:115-142, gma/update.py:127-160
_UPDATE_BLOCKS: Dict[Tuple[str, str], Callable[[int], UpdateSpec]] = {
    ("ptlflow.models.raft.update", "BasicUpdateBlock"): lambda cc: _with_corr_channels(basic_spec(), cc),
    ("ptlflow.models.raft.update", "SmallUpdateBlock"): lambda cc: _with_corr_channels(small_spec(), cc),
    ("ptlflow.models.gma.update", "GMAUpdateBlock"): lambda cc: _with_corr_channels(gma_spec(), cc),
    # ccmr/update.py:110-168: RAFT's encoder / SepConvGRU(512) / heads around an XCiT aggregator that stays the reference's module
    ("ptlflow.models.ccmr.update", "BasicUpdateBlock"): lambda cc: ccmr_spec(cc),
    # ms_raft_plus/update.py:119-153 (stack_coords=False): RAFT's block with a x2 mask head
    ("ptlflow.models.ms_raft_plus.update", "BasicUpdateBlock"): lambda cc: ms_raft_plus_spec(cc),
    # lcv/update.py is raft/update.py reformatted (LCV-RAFT swaps the cost volume for a learnable one, lcv/corr_lcv.py, and
    # keeps RAFT's encoders, update block and loop: lcv_raft.py:143-179)
    ("ptlflow.models.lcv.update", "BasicUpdateBlock"): lambda cc: _with_corr_channels(basic_spec(), cc),
    ("ptlflow.models.lcv.update", "SmallUpdateBlock"): lambda cc: _with_corr_channels(small_spec(), cc),
    # llaflow/update.py: RAFT's block (`llaflow_raft`) and GMA's with heads = 1 (`llaflow`; llaflow/gma.py's Aggregate is
    # gma/gma_utils.py's), around the family's own cost volume (llaflow/corr.py, built directly: no B1 there)
    ("ptlflow.models.llaflow.update", "BasicUpdateBlock"): lambda cc: _with_corr_channels(basic_spec(), cc),
    ("ptlflow.models.llaflow.update", "GMAUpdateBlock"): lambda cc: _with_corr_channels(gma_spec(), cc),
}
# parameters of a matched block that belong to a sub-module the wrapper keeps calling as is (not part of the shape check)
_FOREIGN_PREFIX = {("ptlflow.models.ccmr.update", "BasicUpdateBlock"): "aggregator."}
# (module, class) of the BasicEncoder implementations that are raft/extractor.py:122-194 verbatim
_ENCODERS = {
    ("ptlflow.models.raft.extractor", "BasicEncoder"),
    ("ptlflow.models.raft.extractor", "SmallEncoder"),     # raft_small: bottleneck blocks, extractor.py:197-267
    ("ptlflow.models.gma.extractor", "BasicEncoder"),
    ("ptlflow.models.lcv.extractor", "BasicEncoder"),       # lcv/extractor.py == raft/extractor.py, byte for byte
    ("ptlflow.models.lcv.extractor", "SmallEncoder"),
    ("ptlflow.models.llaflow.extractor", "BasicEncoder"),   # likewise identical to raft/extractor.py
}
# families whose CorrBlock pyramid is not the avg-pool one (sea_raft/corr.py:71-84)
_PYRAMID = {"ptlflow.models.sea_raft.sea_raft": "bilinear_f2"}


def _with_corr_channels(spec: UpdateSpec, corr_channels: int) -> UpdateSpec:
    from dataclasses import replace
    return replace(spec, corr_channels=corr_channels)


def register_update_block(module: str, cls: str, factory: Callable[[int], UpdateSpec]) -> None:
    """Declare another family's update block to be one of the three implementations above (shape check still applies)."""
    _UPDATE_BLOCKS[(module, cls)] = factory


def register_encoder(module: str, cls: str = "BasicEncoder") -> None:
    _ENCODERS.add((module, cls))


def match_update_block(block: torch.nn.Module) -> Optional[UpdateSpec]:
    """The `UpdateSpec` this block implements, or None when it is not (provably) one of ours."""
    from .synth import update_block_shapes
    factory = _UPDATE_BLOCKS.get((type(block).__module__, type(block).__name__))
    if factory is None:
        return None
    foreign = _FOREIGN_PREFIX.get((type(block).__module__, type(block).__name__))
    sd = {k: tuple(v.shape) for k, v in block.state_dict().items() if not (foreign and k.startswith(foreign))}
    w = sd.get("encoder.convc1.weight")
    if w is None or len(w) != 4:
        return None
    spec = factory(int(w[1]))
    mk = sd.get("mask.2.weight")
    if mk is not None and spec.has_mask and mk[0] != spec.mask_channels and len(mk) == 4:
        from dataclasses import replace
        spec = replace(spec, mask_channels=int(mk[0]))      # 9 * scale^2: the upsampling factor is the model's choice
    want = update_block_shapes(spec)
    # GMA registers its unused relative-position tables nowhere under update_block; any extra or missing key is a mismatch
    return spec if sd == want else None


def _basic_encoder_shapes(out_dim: int, norm_fn: str, small: bool = False) -> Dict[str, tuple]:
    """state_dict names/shapes of BasicEncoder (raft/extractor.py:122-170) / SmallEncoder (:197-236) for norm_fn in
    {instance, batch, none}."""
    sh: Dict[str, tuple] = {}

    def conv(name, co, ci, k):
        sh[name + ".weight"] = (co, ci, k, k)
        sh[name + ".bias"] = (co,)

    def norm(name, c):
        if norm_fn == "batch":
            sh[name + ".weight"] = (c,)
            sh[name + ".bias"] = (c,)
            sh[name + ".running_mean"] = (c,)
            sh[name + ".running_var"] = (c,)
            sh[name + ".num_batches_tracked"] = ()

    dims = (32, 32, 64, 96) if small else (64, 64, 96, 128)
    conv("conv1", dims[0], 3, 7)
    norm("norm1", dims[0])
    cin = dims[0]
    for i, (dim, stride) in enumerate(zip(dims[1:], (1, 2, 2)), start=1):
        for j, (ci, s) in enumerate(((cin, stride), (dim, 1))):
            p = f"layer{i}.{j}"
            if small:    # BottleneckBlock: 1x1 -> 3x3 -> 1x1 at a quarter of the width, norm4 on the downsample path
                conv(p + ".conv1", dim // 4, ci, 1)
                conv(p + ".conv2", dim // 4, dim // 4, 3)
                conv(p + ".conv3", dim, dim // 4, 1)
                norm(p + ".norm1", dim // 4)
                norm(p + ".norm2", dim // 4)
                norm(p + ".norm3", dim)
                extra = ".norm4"
            else:
                conv(p + ".conv1", dim, ci, 3)
                conv(p + ".conv2", dim, dim, 3)
                norm(p + ".norm1", dim)
                norm(p + ".norm2", dim)
                extra = ".norm3"
            if s != 1:
                norm(p + extra, dim)
                conv(p + ".downsample.0", dim, ci, 1)
                norm(p + ".downsample.1", dim)
        cin = dim
    conv("conv2", out_dim, dims[3], 1)
    return sh


def match_encoder(enc: torch.nn.Module) -> bool:
    if (type(enc).__module__, type(enc).__name__) not in _ENCODERS:
        return False
    norm_fn = getattr(enc, "norm_fn", None)
    if norm_fn not in ("instance", "batch", "none") or getattr(enc, "dropout", None) is not None:
        return False
    sd = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    out = sd.get("conv2.weight")
    return out is not None and sd == _basic_encoder_shapes(out[0], norm_fn, type(enc).__name__ == "SmallEncoder")


def _supported_envelope(fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int, radius: int) -> bool:
    """What `CorrBlock` (K1-K3) was built for; anything else stays on the reference implementation."""
    if fmap1.dim() != 4 or fmap2.dim() != 4 or fmap1.shape[0] != fmap2.shape[0] or fmap1.shape[1] != fmap2.shape[1]:
        return False
    B, D, h, w = fmap1.shape
    h2, w2 = fmap2.shape[-2:]
    if not (1 <= radius <= 4 and 1 <= num_levels <= 8 and D % 32 == 0):
        return False
    if fmap1.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        return False
    # kernels address one batch element's feature matrix / one source with 32-bit byte offsets
    return max(h * w, h2 * w2) * D * 4 < 2 ** 31 - 1 and B * h * w * num_levels * (2 * radius + 1) ** 2 * 4 < 2 ** 31 - 1


# the accelerated model instance whose forward is running publishes its lookup layout here (None: no such forward on this context)
_LAYOUT: "contextvars.ContextVar[Optional[bool]]" = contextvars.ContextVar("pfk_lookup_channels_last", default=None)
_LAYOUT_HOOKS = "_pfk_layout_hooks"


def _bracket_forward_with_layout(model: torch.nn.Module, channels_last: bool) -> None:
    """forward pre-hook / hook pair on the INSTANCE: sets `_LAYOUT` for the duration of its forward (re-entrant: a token stack)."""
    state = model.__dict__.get(_LAYOUT_HOOKS)
    if state is not None:
        state["channels_last"] = channels_last
        return
    state = {"channels_last": channels_last, "tokens": []}

    def pre(_m, _args, _kwargs=None):
        state["tokens"].append(_LAYOUT.set(state["channels_last"]))

    def post(_m, _args, _out):
        if state["tokens"]:
            _LAYOUT.reset(state["tokens"].pop())

    state["handles"] = (model.register_forward_pre_hook(pre), model.register_forward_hook(post, always_call=True))
    model.__dict__[_LAYOUT_HOOKS] = state


def _make_corr_hook(module_name: str, original):
    pyramid = _PYRAMID.get(module_name, "avgpool")

    def get_corr_block(fmap1, fmap2, num_levels: int = 4, radius: int = 4, alternate_corr: bool = False, **kw):
        # GPU tensors inside the kernels' envelope go to libpfk (inference and — with gradients flowing to the feature maps
        # through `pfk_corr_lookup_bwd_f32` / `pfk_corr_volume_bwd_f32` — training); CPU tensors, alternate_corr, other
        # shapes stay on the reference's own implementation.
        if fmap1.is_cuda and not alternate_corr and not kw and _supported_envelope(fmap1, fmap2, num_levels, radius):
            # layout per model INSTANCE (several instances of one family module may be accelerated with different `update_block=`
            # settings): `accelerate` brackets the instance's forward with hooks that publish its setting in a context variable
            # (`_LAYOUT`); a caller outside such a forward gets the hook's default, set by the last `accelerate` on this module
            cl = _LAYOUT.get()
            if cl is None:
                cl = get_corr_block.channels_last
            return _pfk_get_corr_block(fmap1, fmap2, num_levels=num_levels, radius=radius, pyramid=pyramid, channels_last=cl)
        return original(fmap1=fmap1, fmap2=fmap2, num_levels=num_levels, radius=radius, alternate_corr=alternate_corr, **kw)

    get_corr_block.pyramid = pyramid
    # layout of the lookups handed back: a channels-last view for this package's own update block (no transpose), plain NCHW
    # for a model whose update block stays torch code (set by `accelerate`, see CorrBlock.__init__)
    get_corr_block.channels_last = True
    return get_corr_block


_UPSAMPLE = "_pfk_original_upsample_flow"


class _UpsampleSeam:
    """Seam B5 — `model.upsample_flow(flow, mask)` (raft.py:112-123; the same method in gma.py:130-141, ccmr.py:128-139):
    8x convex upsampling, called once per iteration by the reference's loop, in the reference five torch kernels over a
    [B, 576, h, w] mask (softmax, unfold, multiply, sum, permute-copy).  The replacement is `pfk_convex_upsample_f32`.

    Dispatch by BEHAVIOUR, not by name: on the first GPU call the model's own method and the kernel are run on a small random
    input; the kernel is used from then on only if they agree to 1e-5 (so a family whose `upsample_flow` does something else —
    another scale, a different neighbourhood — keeps its own code).  Gradient graphs, CPU tensors, other mask widths and
    dtypes always take the original."""

    def __init__(self, original):
        self.original = original
        self.ok = None          # None: not probed yet; True / False: the probe's verdict
        self.skip = None        # `_DeadWorkSkip` shared with seam B3 (accelerate(..., skip_dead_upsample=True)), else None

    @staticmethod
    def _kernel(flow: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        B, _, H, W = flow.shape
        pm = mask.permute(0, 2, 3, 1)
        pm = pm.reshape(B * H * W, 576) if pm.is_contiguous() else pm.contiguous().view(B * H * W, 576)   # PfkUpdateBlock's mask: a view
        out = torch.empty(B, 2, 8 * H, 8 * W, device=flow.device, dtype=torch.float32)
        torch.ops.pfk.convex_upsample(flow.contiguous(), pm, out)
        return out

    def _probe(self, device) -> bool:
        g = torch.Generator().manual_seed(5)
        flow = (torch.randn(2, 2, 5, 7, generator=g) * 3).to(device)
        mask = torch.randn(2, 576, 5, 7, generator=g).to(device)
        try:
            with torch.no_grad():
                want = self.original(flow, mask)
            got = self._kernel(flow, mask)
            ok = tuple(want.shape) == tuple(got.shape) and \
                float((want.float() - got).abs().max()) <= 1e-5 * max(1.0, float(want.float().abs().max()))
        except Exception:
            ok = False
        if not ok:
            import warnings
            warnings.warn("ptlflow_amd: this model's upsample_flow is not RAFT's 8x convex upsampling; seam B5 keeps the "
                          "model's own method", RuntimeWarning, stacklevel=3)
        return ok

    def __call__(self, *args, **kwargs):
        # Other families give `upsample_flow` more arguments — ccmr.py:213 / ms_raft_plus.py:199 pass `scale=`, dip a `rate`,
        # rapidflow / dpflow a `factor`, gmflow (feature, bilinear, upsample_factor): anything but the plain two-tensor
        # positional call is the model's own business and goes to its own method exactly as it was written.
        if kwargs or len(args) != 2 or not (torch.is_tensor(args[0]) and torch.is_tensor(args[1])):
            return self.original(*args, **kwargs)
        flow, mask = args
        eligible = (flow.is_cuda and flow.dtype == torch.float32 and mask.dtype == torch.float32 and mask.dim() == 4 and
                    mask.shape[1] == 576 and flow.dim() == 4 and flow.shape[1] == 2 and
                    not (torch.is_grad_enabled() and (flow.requires_grad or mask.requires_grad)))
        if not eligible:
            return self.original(flow, mask)
        if self.ok is None:
            if torch.cuda.is_current_stream_capturing():     # the probe copies host data: illegal inside a graph capture
                return self.original(flow, mask)
            self.ok = self._probe(flow.device)
        if not self.ok:
            return self.original(flow, mask)
        if self.skip is not None:
            if self.skip.upsample_is_dead():
                return self.skip.scratch(flow)      # a non-final iteration's prediction: eval drops it (raft/raft.py:186-192)
            eng = self.skip.take_deferred(mask)
            if eng is not None:
                # seam B3 left mask conv2 to this call: raft/update.py:152 + raft/raft.py:112-123 as ONE kernel on the mask head's
                # hidden activation (K13, bit-identical to the pair); the flow goes pixel-major into the engine's flow slice, which
                # the next block call rewrites from its own `flow` argument anyway
                B, _, H, W = flow.shape
                out = torch.empty(B, 2, 8 * H, 8 * W, device=flow.device, dtype=torch.float32)
                torch.ops.pfk.nchw_to_pm(flow.contiguous(), eng.flow_view)
                eng.mask_upsample(out)
                return out
        return self._kernel(flow, mask)


# (module, class) of correlation MODULES whose volume is a bilinear form of the two feature maps: lcv/corr_lcv.py:19-50
_BILINEAR_VOLUMES = {("ptlflow.models.lcv.corr_lcv", "LearnableCorrBlock")}


class _VolumeToken:
    """What the adapted `compute_cost_volume` hands the caller's loop in place of the list of pyramid levels."""

    def __init__(self, block):
        self.block = block


class _LearnableVolumeSeam:
    """Seam B1 for LCV-RAFT's learnable cost volume (lcv/corr_lcv.py, called from lcv_raft.py:147,167).

    The volume is `(fmap1ᵀ W) fmap2 / √D` with `W = Pᵀ D P` rebuilt from the module's parameters on every call (:20-31), its
    pyramid and radius-r lookup are RAFT's (:44-49, :52-76).  So it IS `CorrBlock(W-transformed fmap1, fmap2)`: the transform is
    one `[B·N, D] x [D, D]` product, K1-K3 do the rest.  Both methods of the module instance are shadowed (the module object,
    its parameters and state_dict keys stay): `compute_cost_volume` returns a token, `forward(token, coords)` looks up.

    Left to the module's own code: CPU tensors, gradient graphs (W is learnable), shapes outside the kernels' envelope, and
    maps so small that the module stops pooling (`min(h, w) <= 2r + 1` at some level: it then repeats a level where RAFT's
    block keeps halving)."""

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.compute_original = module.compute_cost_volume
        self.forward_original = module.forward
        self.channels_last = True

    def eligible(self, fmap1: torch.Tensor, fmap2: torch.Tensor) -> bool:
        m = self.module
        if not (fmap1.is_cuda and fmap1.dim() == 4 and fmap1.shape == fmap2.shape):
            return False
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad or any(p.requires_grad for p in m.parameters())):
            return False
        h, w = fmap1.shape[-2:]
        if any(min(h >> i, w >> i) <= 2 * m.radius + 1 for i in range(m.num_levels - 1)):
            return False
        return _supported_envelope(fmap1, fmap2, m.num_levels, m.radius)

    def compute_cost_volume(self, fmap1, fmap2):
        if not self.eligible(fmap1, fmap2):
            return self.compute_original(fmap1, fmap2)
        B, D, h, w = fmap1.shape
        # W by the module's own arithmetic: its method on a one-pixel crop (:20-31 run in full, the volume part on 1 x 1 maps)
        self.compute_original(fmap1[:, :, :1, :1].contiguous(), fmap2[:, :, :1, :1].contiguous())
        f1w = torch.matmul(fmap1.float().flatten(2).transpose(1, 2), self.module.W.float())       # [B, N, D]: already pixel-major
        block = _pfk_get_corr_block(f1w.view(B, h, w, D).permute(0, 3, 1, 2), fmap2.float(), num_levels=self.module.num_levels,
                                    radius=self.module.radius, channels_last=self.channels_last)
        return _VolumeToken(block)

    def forward(self, corr_pyramid, coords):
        if isinstance(corr_pyramid, _VolumeToken):
            return corr_pyramid.block(coords)
        return self.forward_original(corr_pyramid, coords)


_VOLUME_SEAM = "_pfk_learnable_volume_seam"


# (module, class) whose `forward` was checked to return, in eval mode, only the LAST iteration's upsampled flow: the loop runs
# `range(self.iters)`, `up_mask` is consumed by `self.upsample_flow` alone, and the intermediate `flow_up` tensors go into a list
# that eval drops (raft/raft.py:169-192, gma/gma.py:189-215).  Only for these the dead mask-head + upsampling work may be skipped.
_LAST_FLOW_ONLY_FORWARDS = {
    ("ptlflow.models.raft.raft", "RAFT"),
    ("ptlflow.models.gma.gma", "GMA"),
}


def register_last_flow_only_forward(module: str, cls: str) -> None:
    """Declare that `module.cls.forward` (and `upsample_flow`) is a loop of that kind — checked by whoever registers it."""
    _LAST_FLOW_ONLY_FORWARDS.add((module, cls))
_SKIP = "_pfk_dead_work_skip"


def _defining_class(model: torch.nn.Module, attr: str):
    for cls in type(model).__mro__:
        if attr in cls.__dict__:
            return cls
    return None


class _DeadWorkSkip:
    """§8 f2 at the seams (opt-in: `accelerate(model, skip_dead_upsample=True)`).  In eval the reference's loop computes the mask
    head and the 8x convex upsampling on every iteration and returns the last one only (raft/raft.py:180-192).  This object is shared
    by seam B3 (`PfkUpdateBlock`) and seam B5 (`_UpsampleSeam`): the block counts its calls since the forward began and, knowing
    `model.iters`, computes the mask half of fh|mask conv1 and mask conv2 on the final call only; `upsample_flow` returns a
    never-handed-out scratch tensor before it.  Active per forward only when the model is in eval mode, no gradient graph is
    recorded and `model.iters` is a positive int; any call at or beyond `iters - 1` computes everything."""

    def __init__(self, model: torch.nn.Module, skip_dead: bool = True, fuse: bool = True):
        import weakref
        self._model = weakref.ref(model)
        self.skip_dead = skip_dead   # False: every iteration keeps its mask head + upsampling (only the fused kernel is wanted)
        self.fuse = fuse             # live iterations: mask conv2 + softmax + upsampling as ONE kernel inside `upsample_flow` (K13)
        self.active = False
        self.iters = 0
        self.call = 0           # index of the update-block call in flight (0-based) within the current forward
        self._dead_now = False  # the call in flight is a non-final one: its mask / upsampled flow are never read
        self._scratch = None
        self._deferred = None   # (engine, data_ptr of the mask view handed out) of a live call whose mask conv2 was left to seam B5

    def begin_forward(self, fp32: bool = True) -> None:
        """`fp32` = the state tensors of this forward are float32.  A half / bf16 model (`model.half()`, validate.py:243-244) gets
        its tensors back as fresh casts, so the block cannot tell the calls of one forward apart by storage: nothing is skipped
        for it (the counter used to restart on every call and the final mask was never computed)."""
        m = self._model()
        it = getattr(m, "iters", None) if m is not None else None
        self.active = bool(fp32 and m is not None and not m.training and not torch.is_grad_enabled()
                           and isinstance(it, int) and not isinstance(it, bool) and it >= 1)
        self.iters = it if self.active else 0
        self.call = 0
        self._dead_now = False
        self._deferred = None

    def next_call(self) -> bool:
        """Called by the block per call (after `begin_forward` on the first); True when this call's mask is dead."""
        dead = self.skip_dead and self.active and not torch.is_grad_enabled() and self.call < self.iters - 1
        self._dead_now = dead
        self.call += 1
        self._deferred = None
        return dead

    def may_defer(self) -> bool:
        """A live call of an armed forward may leave mask conv2 to `upsample_flow` (the fused kernel): same conditions as skipping."""
        m = self._model()
        return bool(self.fuse and self.active and not self._dead_now and m is not None and not m.training and not torch.is_grad_enabled())

    def defer(self, engine, mask_view: torch.Tensor) -> None:
        self._deferred = (engine, mask_view.data_ptr())

    def take_deferred(self, mask: torch.Tensor):
        """The engine whose mask conv2 is pending, if `mask` is the (unfilled) view the block handed out for this very call."""
        d, self._deferred = self._deferred, None
        if d is not None and torch.is_tensor(mask) and mask.data_ptr() == d[1]:
            return d[0]
        return None

    def upsample_is_dead(self) -> bool:
        m = self._model()
        return bool(self.active and self._dead_now and m is not None and not m.training and not torch.is_grad_enabled())

    def scratch(self, flow: torch.Tensor) -> torch.Tensor:
        B, _, H, W = flow.shape
        shape = (B, 2, 8 * H, 8 * W)
        if self._scratch is None or tuple(self._scratch.shape) != shape or self._scratch.device != flow.device:
            self._scratch = torch.zeros(shape, device=flow.device, dtype=torch.float32)
        return self._scratch


def _dead_work_skip_for(model: torch.nn.Module, warn: bool = True, skip_dead: bool = True, fuse: bool = True):
    """The shared skip state, or None (with a warning saying why, when the caller asked for it explicitly) when skipping is not
    provably output-preserving for this model."""
    import warnings
    why = None
    fwd, ups = _defining_class(model, "forward"), _defining_class(model, "upsample_flow")
    it = getattr(model, "iters", None)
    if fwd is None or (fwd.__module__, fwd.__name__) not in _LAST_FLOW_ONLY_FORWARDS:
        why = (f"`forward` is defined by {getattr(fwd, '__module__', '?')}.{getattr(fwd, '__name__', '?')}, not by one of the loops checked "
               "to return only the last prediction in eval")
    elif ups is None or (ups.__module__, ups.__name__) not in _LAST_FLOW_ONLY_FORWARDS:
        why = "`upsample_flow` is overridden"
    elif not isinstance(it, int) or isinstance(it, bool) or it < 1:
        why = f"`model.iters` is not a positive int attribute ({it!r})"
    elif model.training:
        why = "the model is in train mode (every prediction enters the loss)"
    elif not (isinstance(model.update_block, PfkUpdateBlock) and model.update_block.spec.has_mask
              and isinstance(model.__dict__.get("upsample_flow"), _UpsampleSeam)):
        why = "seams B3 (a mask-head update block) and B5 are not both installed"
    if why is not None:
        if warn:
            warnings.warn(f"ptlflow_amd: skip_dead_upsample refused, every iteration keeps its mask head + upsampling: {why}",
                          RuntimeWarning, stacklevel=3)
        return None
    return _DeadWorkSkip(model, skip_dead=skip_dead, fuse=fuse)


def accelerate(model: torch.nn.Module, corr: bool = True, update_block: bool = True,
               conv_precision: str = "fp32", encoders: bool = True, upsample: bool = True,
               skip_dead_upsample: Optional[bool] = None, fuse_mask_upsample: Optional[bool] = None) -> torch.nn.Module:
    """Patch seams B1/B3/B4 (+ B5, the model's `upsample_flow` method) of a ptlflow model instance in place and return it.

    ``conv_precision``: "fp32" (default, the parity path) or a split-bf16 mode of the convolutions
    ("bf16x6", "bf16x3", "bf16"), see ``UpdateEngine``.
    ``skip_dead_upsample`` (eval + no_grad + float32 forwards only): compute the mask head and the convex upsampling on the LAST
    iteration only — the reference's loop computes them 32 times and returns the last (raft/raft.py:180-192) — and run that
    last one as the fused mask-conv2 + softmax + upsampling kernel inside `upsample_flow` (K13).  `flows` stays bit-identical.
    ``None`` (default): ON wherever it is provably output-preserving — the model's `forward` and `upsample_flow` are those of
    the loops checked to drop the intermediate predictions (`raft.RAFT`, `gma.GMA`), `model.iters` is a positive int and the
    model is in eval mode when it is accelerated — silently off otherwise; ``True``: the same, with a warning saying why when it
    is refused; ``False`` (opt-out): every iteration keeps its mask head and upsampling, as the reference (the fused kernel
    still serves them where the same checks pass; ``fuse_mask_upsample=False`` keeps the two separate launches).  Re-armed per
    forward: train mode, a gradient graph, a half / bf16 model or a changed `iters` are honoured (`_DeadWorkSkip`)."""
    load_native()
    mod_name = type(model).__module__
    # registered classes (`class raft(RAFT)`) live in the same module as the implementation
    mod = sys.modules.get(mod_name) or importlib.import_module(mod_name)
    if corr and hasattr(mod, "get_corr_block") and not hasattr(mod, _ORIG):
        setattr(mod, _ORIG, mod.get_corr_block)
        mod.get_corr_block = _make_corr_hook(mod_name, getattr(mod, _ORIG))
    if update_block and hasattr(model, "update_block") and not isinstance(model.update_block, PfkUpdateBlock):
        spec = match_update_block(model.update_block)
        if spec is not None:
            model.update_block = PfkUpdateBlock(model.update_block, spec, conv_precision)
    hook = getattr(mod, "get_corr_block", None)
    if corr and hasattr(mod, _ORIG) and hasattr(hook, "channels_last"):      # (default for callers that are not a model's forward)
        hook.channels_last = isinstance(getattr(model, "update_block", None), PfkUpdateBlock)
        _bracket_forward_with_layout(model, hook.channels_last)
    cb = getattr(model, "corr_block", None)
    if corr and isinstance(cb, torch.nn.Module) and (type(cb).__module__, type(cb).__name__) in _BILINEAR_VOLUMES \
            and _VOLUME_SEAM not in cb.__dict__:
        seam = _LearnableVolumeSeam(cb)
        seam.channels_last = isinstance(getattr(model, "update_block", None), PfkUpdateBlock)
        # instance attributes shadow the class's methods (nn.Module.__call__ looks `forward` up on the instance)
        cb.__dict__[_VOLUME_SEAM] = seam
        cb.__dict__["compute_cost_volume"] = seam.compute_cost_volume
        cb.__dict__["forward"] = seam.forward
    if encoders:
        # seam B4: the BasicEncoder feature / context networks (raft, gma: `self.fnet`, `self.cnet`)
        from .encoder import PfkEncoder
        for attr in ("fnet", "cnet"):
            enc = getattr(model, attr, None)
            if enc is not None and not isinstance(enc, PfkEncoder) and match_encoder(enc):
                setattr(model, attr, PfkEncoder(enc, conv_precision, small=type(enc).__name__ == "SmallEncoder"))
    if upsample and callable(getattr(model, "upsample_flow", None)) and _UPSAMPLE not in model.__dict__:
        # an instance attribute shadows the class's method; nn.Module.__setattr__ stores plain callables in __dict__
        model.__dict__[_UPSAMPLE] = model.upsample_flow
        model.__dict__["upsample_flow"] = _UpsampleSeam(model.upsample_flow)
    both = isinstance(getattr(model, "update_block", None), PfkUpdateBlock) and isinstance(model.__dict__.get("upsample_flow"), _UpsampleSeam)
    if _SKIP not in model.__dict__ and (skip_dead_upsample is True or both) \
            and not (skip_dead_upsample is False and fuse_mask_upsample is False):
        skip = _dead_work_skip_for(model, warn=skip_dead_upsample is True, skip_dead=skip_dead_upsample is not False,
                                   fuse=fuse_mask_upsample is not False)
        if skip is not None:
            model.__dict__[_SKIP] = skip
            model.update_block._skip = skip
            model.__dict__["upsample_flow"].skip = skip
    return model


def restore(model: torch.nn.Module) -> torch.nn.Module:
    state = model.__dict__.pop(_LAYOUT_HOOKS, None)
    if state is not None:
        for h in state["handles"]:
            h.remove()
    if _SKIP in model.__dict__:
        del model.__dict__[_SKIP]
        if isinstance(getattr(model, "update_block", None), PfkUpdateBlock):
            model.update_block._skip = None
    if _UPSAMPLE in model.__dict__:
        del model.__dict__[_UPSAMPLE]
        model.__dict__.pop("upsample_flow", None)
    mod = sys.modules.get(type(model).__module__)
    if mod is not None and hasattr(mod, _ORIG):
        mod.get_corr_block = getattr(mod, _ORIG)
        delattr(mod, _ORIG)
    cb = getattr(model, "corr_block", None)
    if cb is not None and _VOLUME_SEAM in getattr(cb, "__dict__", {}):
        for name in (_VOLUME_SEAM, "compute_cost_volume", "forward"):
            cb.__dict__.pop(name, None)
    ub = getattr(model, "update_block", None)
    if isinstance(ub, PfkUpdateBlock):
        model.update_block = ub._ref[0]
    from .encoder import PfkEncoder
    for attr in ("fnet", "cnet"):
        enc = getattr(model, attr, None)
        if isinstance(enc, PfkEncoder):
            setattr(model, attr, enc._ref[0])
    return model
