"""BasicEncoder / SmallEncoder (the feature / context networks) on libpfk kernels — SURVEY.md §8 f3.

Reference: ptlflow/models/raft/extractor.py:122-194 (``BasicEncoder``: 7x7/2 stem, three pairs of ``ResidualBlock``s at
1/2, 1/4, 1/8 resolution, 1x1 output conv; ``norm_fn`` = ``instance`` for fnet, ``batch`` for cnet; gma/ and the other
RAFT descendants reuse it) and :197-267 (``SmallEncoder`` of raft_small: the same skeleton at widths 32/64/96 with
``BottleneckBlock``s — 1x1 -> 3x3 (stride) -> 1x1, :62-119).  Activations stay pixel-major ``[B*H*W, C]`` between kernels:

* stem                     -> ``pfk_conv_stem_f32`` (reads the NCHW image)
* 3x3 / 1x1, stride 1 / 2  -> ``pfk_conv2d_f32`` or ``pfk_conv2d_bf16s`` (implicit GEMM; bias, relu, residual add and the
  block's final relu fused in the epilogue)
* InstanceNorm             -> ``pfk_instnorm_stats_f32`` + ``pfk_norm_apply_f32`` (norm + relu + residual + relu)
* BatchNorm (eval)         -> folded into the convolution's weights and bias at pack time (no kernel)

The tensor handed back is an NCHW-shaped channels-last view of the last pixel-major buffer, which ``CorrBlock`` and
``UpdateEngine.load_state`` consume without a transpose.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import load_native
from .packing import pack_conv_weight, split_bf16_planes
from .update import CONV_PRECISIONS, EPI_LINEAR

# (name, cin, cout, stride) of the six residual blocks, extractor.py:150-153
_BLOCKS = (("layer1.0", 64, 64, 1), ("layer1.1", 64, 64, 1), ("layer2.0", 64, 96, 2), ("layer2.1", 96, 96, 1),
           ("layer3.0", 96, 128, 2), ("layer3.1", 128, 128, 1))
# the six bottleneck blocks of SmallEncoder, extractor.py:216-219
_SMALL_BLOCKS = (("layer1.0", 32, 32, 1), ("layer1.1", 32, 32, 1), ("layer2.0", 32, 64, 2), ("layer2.1", 64, 64, 1),
                 ("layer3.0", 64, 96, 2), ("layer3.1", 96, 96, 1))
EPS = 1e-5


def _out(n: int, s: int) -> int:
    return (n - 1) // s + 1


class EncoderEngine:
    def __init__(self, params: Dict[str, torch.Tensor], norm: str, device: torch.device, conv_precision: str = "fp32",
                 small: bool = False):
        load_native()
        self.small = small
        self.blocks = _SMALL_BLOCKS if small else _BLOCKS
        if norm not in ("instance", "batch", "none"):
            raise ValueError(norm)
        if conv_precision not in CONV_PRECISIONS:
            raise ValueError(f"conv_precision must be one of {sorted(CONV_PRECISIONS)}, got {conv_precision!r}")
        self.ops = torch.ops.pfk
        self.norm = norm
        self.device = device
        self.nsplit = CONV_PRECISIONS[conv_precision]
        # "bf16": K8b (`pfk_conv2d_b16`, UpdateEngine's docstring) — bf16 activation storage between the layers, instance norms
        # included (statistics and normalisation in fp32 arithmetic on the bf16 rows, as F.instance_norm under autocast)
        self.b16 = conv_precision == "bf16"
        self._ws: Optional[torch.Tensor] = None
        self.max_matrix_bytes = 2 ** 31 - 1   # 32-bit byte offsets in the convolution kernels
        self.pack(params)

    # ------------------------------------------------------------------ weights
    def _fold(self, P, conv: str, norm_name: Optional[str]) -> Tuple[torch.Tensor, torch.Tensor]:
        """conv weight / bias with eval-mode BatchNorm folded in (extractor.py:138-139: y = (x - mean)/sqrt(var+eps)*g + b)."""
        dev = self.device
        w = P[conv + ".weight"].detach().to(device=dev, dtype=torch.float32)
        b = P[conv + ".bias"].detach().to(device=dev, dtype=torch.float32)
        if self.norm == "batch" and norm_name is not None:
            g, beta, mean, var = (P[f"{norm_name}.{k}"].detach().to(device=dev, dtype=torch.float32)
                                  for k in ("weight", "bias", "running_mean", "running_var"))
            scale = g / torch.sqrt(var + EPS)
            w = w * scale[:, None, None, None]
            b = (b - mean) * scale + beta
        return w, b.contiguous()

    def _pk(self, w: torch.Tensor) -> torch.Tensor:
        cin = w.shape[1]
        if self.b16:
            return pack_conv_weight(w, [(0, cin, cin)], kpad=64).to(torch.bfloat16).contiguous()
        packed = pack_conv_weight(w, [(0, cin, cin)])
        return packed if self.nsplit == 0 else split_bf16_planes(packed, self.nsplit)

    def pack(self, P: Dict[str, torch.Tensor]) -> None:
        W: Dict[str, torch.Tensor] = {}
        w, b = self._fold(P, "conv1", "norm1")
        assert w.shape[1:] == (3, 7, 7), "stem must be Conv2d(3, C, 7, stride=2, padding=3)"
        W["stem.w"] = w.permute(2, 3, 1, 0).reshape(49, 3, w.shape[0]).contiguous()
        W["stem.b"] = b
        convs = (("conv1", "norm1"), ("conv2", "norm2"), ("conv3", "norm3")) if self.small else (("conv1", "norm1"), ("conv2", "norm2"))
        for name, cin, cout, stride in self.blocks:
            for conv, nrm in convs:
                w, b = self._fold(P, f"{name}.{conv}", f"{name}.{nrm}")
                W[f"{name}.{conv}.w"], W[f"{name}.{conv}.b"] = self._pk(w), b
            if stride != 1:
                w, b = self._fold(P, f"{name}.downsample.0", f"{name}.downsample.1")
                W[f"{name}.ds.w"], W[f"{name}.ds.b"] = self._pk(w), b
        w, b = self._fold(P, "conv2", None)
        W["out.w"], W["out.b"] = self._pk(w), b
        self.out_dim = w.shape[0]
        self.stem_dim = W["stem.b"].numel()
        self.w = W

    # ------------------------------------------------------------------ kernels
    def _conv(self, x, B, H, W, k, key, cout, stride=1, relu=False, residual=None, relu2=False, fp32_out=None):
        M = B * _out(H, stride) * _out(W, stride)
        if self.b16:
            # bf16 rows out, also in front of an instance norm: under the reference's autocast switch a convolution returns a
            # 16-bit tensor and F.instance_norm (fp32 arithmetic) reads that — statistics and normalisation of the rounded values
            # (half the bytes of the three passes over the largest activations of the forward); fp32 only for the network's output
            out = torch.empty(M, cout, device=self.device, dtype=torch.float32 if fp32_out else torch.bfloat16)
            self.ops.conv2d_b16([x], B, H, W, k, k, self.w[key + ".w"], self.w[key + ".b"], cout, EPI_LINEAR, relu, 1.0, out,
                                None, None, None, None, residual, stride, relu2)
            return out
        out = torch.empty(M, cout, device=self.device, dtype=torch.float32)
        self.ops.conv2d([x], B, H, W, k, k, self.w[key + ".w"], self.w[key + ".b"], cout, EPI_LINEAR, relu, 1.0, out,
                        None, None, None, None, residual, stride, relu2)
        return out

    def _inorm(self, x, B, HW, relu, residual=None, relu2=False):
        """InstanceNorm2d (+relu, + residual add + relu) in place on the pixel-major buffer x — fp32 rows, or the bf16 rows of the K8b
        path (only the stem, a VALU / fp32-MFMA kernel, hands over fp32 rows there: those are normalised into a fresh bf16 buffer)."""
        C = x.shape[1]
        need = self.ops.instnorm_workspace_bytes(B, C)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device, dtype=torch.uint8)
        mean = torch.empty(B * C, device=self.device, dtype=torch.float32)
        rstd = torch.empty(B * C, device=self.device, dtype=torch.float32)
        self.ops.instnorm_stats(x, B, HW, EPS, mean, rstd, self._ws)
        out = torch.empty(x.shape[0], C, device=self.device, dtype=torch.bfloat16) if (self.b16 and x.dtype != torch.bfloat16) else x
        self.ops.norm_apply(x, mean, rstd, residual, out, B, HW, relu, relu2)      # (bf16 rows: in place, like the fp32 path)
        return out

    def _bottleneck(self, x, B, H, W, name, cout, stride):
        """BottleneckBlock.forward (extractor.py:110-119): relu(norm(1x1)) -> relu(norm(3x3, stride)) -> relu(norm(1x1)); relu(x + y)."""
        Ho, Wo = _out(H, stride), _out(W, stride)
        mid = cout // 4
        if self.norm == "instance":
            y = self._inorm(self._conv(x, B, H, W, 1, f"{name}.conv1", mid), B, H * W, relu=True)
            y = self._inorm(self._conv(y, B, H, W, 3, f"{name}.conv2", mid, stride), B, Ho * Wo, relu=True)
            y = self._conv(y, B, Ho, Wo, 1, f"{name}.conv3", cout)
            if stride != 1:
                x = self._inorm(self._conv(x, B, H, W, 1, f"{name}.ds", cout, stride), B, Ho * Wo, relu=False)
            return self._inorm(y, B, Ho * Wo, relu=True, residual=x, relu2=True)
        y = self._conv(x, B, H, W, 1, f"{name}.conv1", mid, relu=True)
        y = self._conv(y, B, H, W, 3, f"{name}.conv2", mid, stride, relu=True)
        if stride != 1:
            x = self._conv(x, B, H, W, 1, f"{name}.ds", cout, stride)
        return self._conv(y, B, Ho, Wo, 1, f"{name}.conv3", cout, relu=True, residual=x, relu2=True)

    def _block(self, x, B, H, W, name, cout, stride):
        if self.small:
            return self._bottleneck(x, B, H, W, name, cout, stride)
        Ho, Wo = _out(H, stride), _out(W, stride)
        if self.norm == "instance":
            y = self._inorm(self._conv(x, B, H, W, 3, f"{name}.conv1", cout, stride), B, Ho * Wo, relu=True)
            y = self._conv(y, B, Ho, Wo, 3, f"{name}.conv2", cout)
            if stride != 1:
                x = self._inorm(self._conv(x, B, H, W, 1, f"{name}.ds", cout, stride), B, Ho * Wo, relu=False)
            return self._inorm(y, B, Ho * Wo, relu=True, residual=x, relu2=True)
        act = True   # "batch" (folded) and "none": relu straight in the epilogue
        y = self._conv(x, B, H, W, 3, f"{name}.conv1", cout, stride, relu=act)
        if stride != 1:
            x = self._conv(x, B, H, W, 1, f"{name}.ds", cout, stride)
        return self._conv(y, B, Ho, Wo, 3, f"{name}.conv2", cout, relu=act, residual=x, relu2=True)

    @torch.no_grad()
    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        """``img`` [B, 3, H, W] fp32 (already normalised / padded) -> [B, out_dim, H/8, W/8] (channels-last strides)."""
        if not img.is_cuda:
            raise RuntimeError("EncoderEngine needs GPU tensors (no CPU fallback)")
        img = img.float().contiguous()
        B, _, H, W = img.shape
        H1, W1 = _out(H, 2), _out(W, 2)
        # the convolution kernels address a source with 32-bit byte offsets: keep every activation matrix under 2 GiB
        per_image = H1 * W1 * 128 * 4   # widest activation row: 128 floats (32 at half resolution for the small encoder; same bound)
        max_b = max(1, self.max_matrix_bytes // per_image)
        if B > max_b:
            return torch.cat([self(img[i:i + max_b]) for i in range(0, B, max_b)], 0)
        # K8b: the MFMA stem kernel rounds to bf16 itself (widths 32 / 64), also in front of an instance norm (`_conv`'s note)
        stem_b16 = self.b16 and self.stem_dim in (32, 64)
        x = torch.empty(B * H1 * W1, self.stem_dim, device=self.device, dtype=torch.bfloat16 if stem_b16 else torch.float32)
        self.ops.conv_stem(img, self.w["stem.w"], self.w["stem.b"], x, self.norm != "instance")
        if self.norm == "instance":
            x = self._inorm(x, B, H1 * W1, relu=True)
        elif self.b16 and not stem_b16:
            x = x.to(torch.bfloat16)
        h, w = H1, W1
        for name, _cin, cout, stride in self.blocks:
            x = self._block(x, B, h, w, name, cout, stride)
            h, w = _out(h, stride), _out(w, stride)
        y = self._conv(x, B, h, w, 1, "out", self.out_dim, fp32_out=True)
        return y.view(B, h, w, self.out_dim).permute(0, 3, 1, 2)


class PfkEncoder(torch.nn.Module):
    """Drop-in for ``model.fnet`` / ``model.cnet`` of a live ptlflow model (``BasicEncoder``, raft/extractor.py:122-194):
    same ``forward(x)`` contract (a tensor, or a list of two images that is batched and split again, :172-193), same
    parameters — the reference sub-modules are re-registered under their own names, so state_dict keys, checkpoints and
    optimizers are untouched.  GPU inference goes to ``EncoderEngine``; training mode, gradient graphs, CPU tensors and
    GroupNorm stay on the reference's own code."""

    def __init__(self, ref: torch.nn.Module, conv_precision: str = "fp32", small: bool = False):
        super().__init__()
        self.small = small
        for name, child in ref.named_children():
            self.add_module(name, child)
        self._ref = [ref]   # in a list: not registered twice in the module tree
        self.training = ref.training
        self.norm_fn = ref.norm_fn
        self.conv_precision = conv_precision
        self._engine: Optional[EncoderEngine] = None
        self._versions = None

    def _get_engine(self, device) -> EncoderEngine:
        ref = self._ref[0]
        v = tuple((t.data_ptr(), t._version) for t in list(ref.parameters()) + list(ref.buffers()))
        if self._engine is None or self._engine.device != device or v != self._versions:
            self._engine = EncoderEngine(ref.state_dict(), self.norm_fn, device, self.conv_precision, self.small)
            self._versions = v
        return self._engine

    def forward(self, x):
        ref = self._ref[0]
        is_list = isinstance(x, (tuple, list))
        probe = x[0] if is_list else x
        needs_graph = torch.is_grad_enabled() and (probe.requires_grad or any(p.requires_grad for p in ref.parameters()))
        ref.training = self.training   # .train() / .eval() reach this wrapper and the shared children, not `ref` itself
        if self.training or needs_graph or not probe.is_cuda or self.norm_fn not in ("instance", "batch", "none"):
            return ref(x)
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(list(x), dim=0)
        y = self._get_engine(x.device)(x)
        if probe.dtype != torch.float32 and probe.is_floating_point():
            y = y.to(probe.dtype)      # a half / bf16 model (`model.half()`, validate.py:243-244) gets its features in its own dtype
        if is_list:
            y = torch.split(y, [batch_dim, batch_dim], dim=0)
        return y
