"""TEST INFRASTRUCTURE — loads the *real* reference (hmorimitsu/ptlflow at /root/reference)
in this container so the oracle restatement can be pinned against it.

Used by `oracle/make_golden.py`, the `reference`-marked tests and bench.py's `dropin` / `cpu_baseline`
legs (the reference as the thing being accelerated / the baseline being timed); nothing under
`ptlflow_amd/` may import it.

`import ptlflow` does not work here (lightning, jsonargparse, loguru, torchmetrics, cv2,
torchvision, timm are absent and there is no network) -- SURVEY.md §8(c).  The hot-path
files themselves only need torch + scipy + einops, so:

  1. stub `lightning.pytorch`, `loguru`, `torchmetrics` with a few lines each;
  2. register *namespace* modules for `ptlflow`, `ptlflow.models`, `ptlflow.models.<family>`,
     `ptlflow.utils`... whose `__path__` points into /root/reference, so the heavy package
     `__init__.py` files are never executed while every leaf module (raft/corr.py,
     raft/update.py, raft/raft.py, utils/correlation.py ...) is the reference's own file,
     imported unmodified from where it lies.

Nothing is copied into the history: the reference code runs from /root/reference where that exists (the build
container) and otherwise from the archive `oracle/stage_ref.py` packed there (`oracle/_ref/`, git-ignored, shipped to the GPU
box with the working tree like the built `.so` files) — the same unmodified files either way.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

def _has_tree(root) -> bool:
    return bool(root) and os.path.isdir(os.path.join(root, "ptlflow", "models", "raft"))


def _resolve_root():
    """$PTLFLOW_REFERENCE_ROOT, else /root/reference, else the archive staged by oracle/stage_ref.py (unpacked once per
    content hash into the temp dir).  Returns (root, kind) with kind in {"env", "tree", "staged", None}."""
    env = os.environ.get("PTLFLOW_REFERENCE_ROOT")
    if env:
        return (env, "env") if _has_tree(env) else (env, None)
    if _has_tree("/root/reference") and not os.environ.get("PFK_REFERENCE_FORCE_STAGED"):   # (the flag: test the staged path here)
        return "/root/reference", "tree"
    try:
        from oracle import stage_ref
        root = stage_ref.unpack()
    except Exception as e:                              # a damaged archive is "no reference" — said out loud, then visible in the skips
        import warnings
        warnings.warn(f"oracle/_ref could not be unpacked ({e!r}): the reference-class tests / bench legs will not run", RuntimeWarning)
        root = None
    return (root, "staged") if _has_tree(root) else ("/root/reference", None)


REFERENCE_ROOT, REFERENCE_KIND = _resolve_root()


def reference_available() -> bool:
    return REFERENCE_KIND is not None


def _install_stubs() -> None:
    import torch.nn as nn

    if "lightning" not in sys.modules:
        lightning = types.ModuleType("lightning")
        pl = types.ModuleType("lightning.pytorch")

        class LightningModule(nn.Module):
            def save_hyperparameters(self, *a, **k):
                pass

            def log(self, *a, **k):
                pass

            def log_dict(self, *a, **k):
                pass

            # lightning's DeviceDtypeModuleMixin: several families read `self.dtype` / `self.device` in forward()
            @property
            def dtype(self):
                return next((p.dtype for p in self.parameters() if p.is_floating_point()), None)

            @property
            def device(self):
                return next((p.device for p in self.parameters()), None)

        pl.LightningModule = LightningModule
        lightning.pytorch = pl
        sys.modules["lightning"] = lightning
        sys.modules["lightning.pytorch"] = pl

    if "loguru" not in sys.modules:
        loguru = types.ModuleType("loguru")

        class _Logger:
            def __getattr__(self, name):
                return lambda *a, **k: None

        loguru.logger = _Logger()
        sys.modules["loguru"] = loguru

    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")

        class Metric(nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

            def add_state(self, name, default, dist_reduce_fx=None):
                self.register_buffer(name, default)
                self.__dict__.setdefault("_pfk_defaults", {})[name] = default.clone()

            def reset(self):
                for name, default in self.__dict__.get("_pfk_defaults", {}).items():
                    getattr(self, name).copy_(default)

            def forward(self, *a, **k):
                # torchmetrics.Metric.forward: the metric of THIS batch (update on cleared states, compute)
                self.reset()
                self.update(*a, **k)
                return self.compute()

        tm.Metric = Metric
        sys.modules["torchmetrics"] = tm

    # ccmr/xcit.py needs four helpers of timm, ccmr|ms_raft_plus/extractor.py import torchvision's functional transforms
    # at module level; neither package is installed here.  Minimal stand-ins (enough to CONSTRUCT those models on CPU
    # for the patch-dispatch tests — nothing on the measured path uses them).
    if "timm" not in sys.modules:
        import collections.abc
        from itertools import repeat

        timm = types.ModuleType("timm")
        tmodels = types.ModuleType("timm.models")
        vit = types.ModuleType("timm.models.vision_transformer")
        layers = types.ModuleType("timm.models.layers")

        class Mlp(nn.Module):
            def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
                super().__init__()
                self.fc1 = nn.Linear(in_features, hidden_features or in_features)
                self.act = act_layer()
                self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
                self.drop = nn.Dropout(drop)

            def forward(self, x):
                return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

        class DropPath(nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__()

            def forward(self, x):
                return x

        def to_2tuple(x):
            return tuple(x) if isinstance(x, collections.abc.Iterable) and not isinstance(x, str) else tuple(repeat(x, 2))

        vit.Mlp = Mlp
        layers.DropPath, layers.trunc_normal_, layers.to_2tuple = DropPath, nn.init.trunc_normal_, to_2tuple
        timm.models, tmodels.vision_transformer, tmodels.layers = tmodels, vit, layers
        sys.modules.update({"timm": timm, "timm.models": tmodels, "timm.models.vision_transformer": vit,
                            "timm.models.layers": layers})
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvf = types.ModuleType("torchvision.transforms.functional")
        tv.transforms, tvt.functional = tvt, tvf

        def resize(img, size, *a, **k):
            # ccmr|ms_raft_plus/extractor.py call TF.resize(feature_map, (h, w)) on tensors: torchvision's tensor path is
            # bilinear interpolate, align_corners=False, antialias on.  The accelerated GPU run and the CPU run it is
            # compared with both go through this stand-in, so it cannot create or hide a difference between them.
            import torch.nn.functional as F
            return F.interpolate(img, size=tuple(size), mode="bilinear", align_corners=False, antialias=True)

        tvf.resize = resize
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})


def _namespace(name: str, path: str) -> None:
    if name in sys.modules:
        return
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    mod.__package__ = name
    sys.modules[name] = mod
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(sys.modules[parent], child, mod)


_LOADED = False


def load() -> None:
    """Make `ptlflow.models.<family>.*` importable from /root/reference (idempotent)."""
    global _LOADED
    if _LOADED:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    _install_stubs()
    root = os.path.join(REFERENCE_ROOT, "ptlflow")
    _namespace("ptlflow", root)
    _namespace("ptlflow.models", os.path.join(root, "models"))
    _namespace("ptlflow.utils", os.path.join(root, "utils"))
    _namespace("ptlflow.utils.external", os.path.join(root, "utils", "external"))
    _namespace("ptlflow.models.base_model", os.path.join(root, "models", "base_model"))
    for fam in ("raft", "gma", "sea_raft", "ccmr", "ms_raft_plus"):
        _namespace(f"ptlflow.models.{fam}", os.path.join(root, "models", fam))
    _LOADED = True


def ensure_family(fam: str) -> bool:
    """Make `ptlflow.models.<fam>` importable too (any family of the reference's zoo, not only the five hot-path ones).
    Returns False where the family's directory is not there (the staged archive carries the hot-path families only)."""
    load()
    path = os.path.join(REFERENCE_ROOT, "ptlflow", "models", fam)
    if not os.path.isdir(path):
        return False
    if f"ptlflow.models.{fam}" not in sys.modules:
        _namespace(f"ptlflow.models.{fam}", path)
    return True


def ref_module(dotted: str):
    """Import a reference module, e.g. ref_module('ptlflow.models.raft.corr')."""
    load()
    return importlib.import_module(dotted)


def build_raft(small: bool = False, seed: int = 1234, **kwargs):
    """Instantiate the reference RAFT / RAFTSmall with seeded default init, eval mode."""
    import torch

    m = ref_module("ptlflow.models.raft.raft")
    torch.manual_seed(seed)
    model = (m.RAFTSmall if small else m.RAFT)(**kwargs)
    return model.eval()


# --------------------------------------------------------------------------------------------------------------------------
# The reference's own ENTRY POINTS: `ptlflow.get_model` (ptlflow/__init__.py:65-125, the registry front end) and the top-level
# script `model_benchmark.py` (its `estimate_inference_time`, :422-466, with `ptlflow.utils.timer.Timer`).  Both files are
# executed unmodified; what is stood in for are the third-party packages this image lacks: `jsonargparse` (the handful of calls
# `get_model` makes to build a model with its default arguments), `plotly.express`, and lightning's CLI / trainer classes
# (base classes of `PTLFlowCLI`, which `model_benchmark.py` imports and only its `main` uses).
# --------------------------------------------------------------------------------------------------------------------------
def _install_script_stubs() -> None:
    import inspect

    if "jsonargparse" not in sys.modules:
        ja = types.ModuleType("jsonargparse")

        class Namespace:
            def __init__(self, **kw):
                self.__dict__.update(kw)

            def __getitem__(self, k):
                return self.__dict__[k]

            def __setitem__(self, k, v):
                self.__dict__[k] = v

            def __contains__(self, k):
                return k in self.__dict__

            def get(self, k, default=None):
                return self.__dict__.get(k, default)

            def keys(self):
                return self.__dict__.keys()

            def items(self):
                return self.__dict__.items()

            def as_dict(self):
                return {k: (v.as_dict() if isinstance(v, Namespace) else v) for k, v in self.__dict__.items()}

            def clone(self):
                return Namespace(**{k: (v.clone() if isinstance(v, Namespace) else v) for k, v in self.__dict__.items()})

            def __repr__(self):
                return "Namespace(" + ", ".join(f"{k}={v!r}" for k, v in self.__dict__.items()) + ")"

        def _defaults(cls):
            """keyword defaults of a class's constructor (jsonargparse.add_class_arguments reads the same signature)"""
            out = {}
            for name, prm in inspect.signature(cls.__init__).parameters.items():
                if name == "self" or prm.kind in (prm.VAR_POSITIONAL, prm.VAR_KEYWORD):
                    continue
                if prm.default is inspect.Parameter.empty:
                    raise TypeError(f"{cls.__name__}.__init__ has a required argument {name!r}: the jsonargparse stand-in only builds defaults")
                out[name] = prm.default
            return out

        class ArgumentParser:
            def __init__(self, *a, **k):
                self._args, self._classes = {}, {}

            def add_argument(self, *names, type=None, default=None, **k):
                self._args[names[-1].lstrip("-").replace("-", "_")] = (type, default)

            def add_class_arguments(self, cls, nested_key=None, **k):
                self._classes[nested_key] = cls

            def parse_args(self, args=None):
                if args:
                    raise NotImplementedError("the jsonargparse stand-in parses no command line")
                ns = Namespace(**{name: default for name, (_t, default) in self._args.items()})
                for key, cls in self._classes.items():
                    ns[key] = Namespace(**_defaults(cls))
                return ns

            def parse_object(self, obj):
                return Namespace(**dict(obj))

            def instantiate_classes(self, cfg):
                out = Namespace()
                for name, value in cfg.items():
                    typ = self._args.get(name, (None, None))[0]
                    if inspect.isclass(typ) and isinstance(value, Namespace):
                        value = typ(**value.as_dict())
                    out[name] = value
                return out

        ja.Namespace, ja.ArgumentParser = Namespace, ArgumentParser
        sys.modules["jsonargparse"] = ja

    if "plotly" not in sys.modules:
        plotly = types.ModuleType("plotly")
        px = types.ModuleType("plotly.express")
        plotly.express = px
        sys.modules.update({"plotly": plotly, "plotly.express": px})

    lightning = sys.modules["lightning"]
    if not hasattr(lightning, "Trainer"):
        pl = sys.modules["lightning.pytorch"]
        lightning.LightningModule = pl.LightningModule
        lightning.LightningDataModule = type("LightningDataModule", (), {})
        lightning.Trainer = type("Trainer", (), {})
        cli = types.ModuleType("lightning.pytorch.cli")
        cli.ArgsType = object
        cli.LightningArgumentParser = type("LightningArgumentParser", (), {})
        cli.LightningCLI = type("LightningCLI", (), {})
        cli.SaveConfigCallback = type("SaveConfigCallback", (), {})
        util = types.ModuleType("lightning.pytorch.utilities")
        rz = types.ModuleType("lightning.pytorch.utilities.rank_zero")
        rz.rank_zero_warn = lambda *a, **k: None
        pl.cli, pl.utilities, util.rank_zero = cli, util, rz
        sys.modules.update({"lightning.pytorch.cli": cli, "lightning.pytorch.utilities": util,
                            "lightning.pytorch.utilities.rank_zero": rz})


_SCRIPTS_LOADED = False


def load_scripts() -> None:
    """After this, `import ptlflow; ptlflow.get_model("raft")` goes through the reference's own `ptlflow/__init__.py` and registry
    (the hot-path families register themselves when their model module is imported: `ptlflow.models.raft.raft`, `...gma.gma`), and
    `ref_script("model_benchmark")` imports the reference's benchmarking script.  Idempotent."""
    global _SCRIPTS_LOADED
    load()
    if _SCRIPTS_LOADED:
        return
    init = os.path.join(REFERENCE_ROOT, "ptlflow", "__init__.py")
    if not os.path.isfile(init):
        raise RuntimeError(f"{init} is not there (an archive staged before the entry points were added?)")
    _install_script_stubs()
    _namespace("ptlflow.utils.lightning", os.path.join(REFERENCE_ROOT, "ptlflow", "utils", "lightning"))
    pkg = sys.modules["ptlflow"]                      # the namespace module: `ptlflow/__init__.py` is executed INTO it, so the
    pkg.__file__ = init                               # sub-package stubs registered by load() stay in place (ptlflow/models/__init__.py,
    with open(init, "rb") as fh:                      # which imports the whole zoo, is never run)
        exec(compile(fh.read(), init, "exec"), pkg.__dict__)
    for mod in ("ptlflow.models.raft.raft", "ptlflow.models.gma.gma"):       # their `@register_model` classes enter the registry
        importlib.import_module(mod)
    _SCRIPTS_LOADED = True


def ref_script(name: str):
    """Import one of the reference's top-level scripts (e.g. `model_benchmark`) as a module, unmodified."""
    import importlib.util
    load_scripts()
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(REFERENCE_ROOT, name + ".py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(name, None)
        raise
    return mod
