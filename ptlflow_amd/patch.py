"""Drop-in shim for a *live* ptlflow model (a machine that has ptlflow + lightning installed).

    import ptlflow, ptlflow_amd.patch
    model = ptlflow.get_model("raft", ckpt_path="things").eval().cuda()
    ptlflow_amd.patch.accelerate(model)          # <- the only new line
    out = model({"images": images})              # validate.py / infer.py / model_benchmark.py unchanged

What gets replaced:
* seam B1 — every model family binds ``get_corr_block`` by name into its own model module at import
  (`from .corr import get_corr_block`, raft.py:10, gma.py, sea_raft.py, ccmr.py, ms_raft_plus.py), so the
  patch target is the *model module's* global, not `corr.py`;
* seam B3 — ``model.update_block`` is wrapped by `PfkUpdateBlock`, which keeps the original sub-modules
  (state_dict keys, checkpoints and optimizers are untouched) and only overrides ``forward``;
* seam B4 — ``model.fnet`` / ``model.cnet`` (`BasicEncoder`) are wrapped by `PfkEncoder` the same way.

Nothing in ptlflow is edited or copied; `restore(model)` undoes the patch.
"""
from __future__ import annotations

import importlib
import sys
from typing import Optional

import torch

from . import load_native
from .corr import get_corr_block as _pfk_get_corr_block
from .update import PfkUpdateBlock, UpdateSpec, basic_spec, gma_spec, small_spec

_ORIG = "_pfk_original_get_corr_block"

# class name of model.update_block -> spec factory (raft/update.py:115-142)
_SPECS = {
    "BasicUpdateBlock": lambda m: basic_spec(getattr(m, "corr_levels", 4), getattr(m, "corr_radius", 4)),
    "GMAUpdateBlock": lambda m: gma_spec(getattr(m, "corr_levels", 4), getattr(m, "corr_radius", 4)),
    "SmallUpdateBlock": lambda m: small_spec(getattr(m, "corr_levels", 4), getattr(m, "corr_radius", 3)),
}
# families whose CorrBlock pyramid is not the avg-pool one
_PYRAMID = {"ptlflow.models.sea_raft.sea_raft": "bilinear_f2"}


def _make_corr_hook(module_name: str, original):
    pyramid = _PYRAMID.get(module_name, "avgpool")

    def get_corr_block(fmap1, fmap2, num_levels: int = 4, radius: int = 4, alternate_corr: bool = False, **kw):
        # GPU fp32 inference goes to the kernels; anything else (CPU tensors, training graphs that need
        # gradients to the feature maps, alternate_corr) stays on the reference's own implementation.
        if (fmap1.is_cuda and not alternate_corr and not (torch.is_grad_enabled() and fmap1.requires_grad)):
            return _pfk_get_corr_block(fmap1, fmap2, num_levels=num_levels, radius=radius, pyramid=pyramid)
        return original(fmap1=fmap1, fmap2=fmap2, num_levels=num_levels, radius=radius, alternate_corr=alternate_corr, **kw)

    return get_corr_block


def accelerate(model: torch.nn.Module, corr: bool = True, update_block: bool = True,
               conv_precision: str = "fp32", encoders: bool = True) -> torch.nn.Module:
    """Patch seams B1/B3 of a ptlflow model instance in place and return it.

    ``conv_precision``: "fp32" (default, the parity path) or a split-bf16 mode of the update block's convolutions
    ("bf16x6", "bf16x3", "bf16"), see ``UpdateEngine``."""
    load_native()
    mod_name = type(model).__module__
    # registered classes (`class raft(RAFT)`) live in the same module as the implementation
    mod = sys.modules.get(mod_name) or importlib.import_module(mod_name)
    if corr and hasattr(mod, "get_corr_block") and not hasattr(mod, _ORIG):
        setattr(mod, _ORIG, mod.get_corr_block)
        mod.get_corr_block = _make_corr_hook(mod_name, getattr(mod, _ORIG))
    if update_block and hasattr(model, "update_block") and not isinstance(model.update_block, PfkUpdateBlock):
        ub = model.update_block
        factory = _SPECS.get(type(ub).__name__)
        if factory is not None:
            spec: UpdateSpec = factory(model)
            model.update_block = PfkUpdateBlock(ub, spec, conv_precision)
    if encoders:
        # seam B4: the BasicEncoder feature / context networks (raft, gma, ... : `self.fnet`, `self.cnet`)
        from .encoder import PfkEncoder
        for attr in ("fnet", "cnet"):
            enc = getattr(model, attr, None)
            if enc is not None and type(enc).__name__ == "BasicEncoder" and getattr(enc, "norm_fn", None) in ("instance", "batch", "none"):
                setattr(model, attr, PfkEncoder(enc, conv_precision))
    return model


def restore(model: torch.nn.Module) -> torch.nn.Module:
    mod = sys.modules.get(type(model).__module__)
    if mod is not None and hasattr(mod, _ORIG):
        mod.get_corr_block = getattr(mod, _ORIG)
        delattr(mod, _ORIG)
    ub = getattr(model, "update_block", None)
    if isinstance(ub, PfkUpdateBlock):
        model.update_block = ub._ref[0]
    from .encoder import PfkEncoder
    for attr in ("fnet", "cnet"):
        enc = getattr(model, attr, None)
        if isinstance(enc, PfkEncoder):
            setattr(model, attr, enc._ref[0])
    return model
