#!/usr/bin/env python3
"""Time the pyramid lookup (K3) and the on-demand correlation (K7) on the north-star shape (GPU box)."""
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")   # tuning script: uses the pfk_debug_set_* knobs
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")
torch.manual_seed(0)
for B in (1, 8):
    h, w, L, r = 55, 128, 4, 4
    N = h * w
    lv, hh, ww = [], h, w
    for l in range(L):
        lv.append(torch.randn(B * N, hh, ww, device=dev)); hh //= 2; ww //= 2
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys], 0)[None] + torch.randn(B, 2, h, w, device=dev) * 6).contiguous()
    out = torch.empty(B * N, 324, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = None
    for variant in (4, 14, 8, 4, 14):     # pixels per workgroup; 14 = 4 with cross-lane tap reads instead of the LDS patch
        ops.debug_set_lookup_pix(variant)
        for _ in range(3):
            ops.corr_lookup(lv, coords, r, out)
        e0.record()
        for _ in range(50):
            ops.corr_lookup(lv, coords, r, out)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 50
        alg = B * (N * L * (100 + 81) * 4 + 8 * N)
        same = True if ref is None else bool(torch.equal(ref, out))
        ref = out.clone() if ref is None else ref
        ops.debug_set_lookup_pix(4)
        print(f"lookup B={B} variant {variant}: {us:.1f} us, algorithmic {alg/1e6:.1f} MB -> {alg/us/1e6:.2f} TB/s ({100*alg/us/1e6/8:.1f}% of 8 TB/s) same={same}")
    # K7 on-demand correlation: the per-pixel kernel (mode 1) against the window-sharing MFMA kernel on 8x4 / 8x8 patches (2 / 3)
    # and the library's choice (0), for a smooth flow field (bicubic-upsampled low-resolution noise, +-20 px: what a flow network
    # produces) and for the iid sigma = 6 px field above (every patch's box overflows: the window-sharing kernel's fallback)
    for (hh2, ww2) in ((h, w), (2 * h, 2 * w)):
        if hh2 != h and B > 1:
            continue
        Nn = hh2 * ww2
        f1 = torch.randn(B, hh2, ww2, 256, device=dev); f2 = torch.randn(B, hh2, ww2, 256, device=dev)
        ys2, xs2 = torch.meshgrid(torch.arange(hh2, device=dev, dtype=torch.float32), torch.arange(ww2, device=dev, dtype=torch.float32), indexing="ij")
        grid = torch.stack([xs2, ys2], 0)[None]
        for amp in (1.5, 4.0):
          smooth = torch.nn.functional.interpolate(torch.randn(B, 2, hh2 // 8 + 2, ww2 // 8 + 2, device=dev) * amp, size=(hh2, ww2), mode="bicubic", align_corners=True)
          noise = torch.randn(B, 2, hh2, ww2, device=dev) * 6
          half = (torch.rand(B, 1, hh2 // 8 + 1, ww2 // 8 + 1, device=dev) < 0.5).float().repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :hh2, :ww2]
          fields = {f"smooth{amp}": grid + smooth} if amp < 2 else {f"smooth{amp}": grid + smooth, "iid-6px": grid + noise,
                                                                     "half-iid": grid + smooth + noise * half}
          for fname, cc in fields.items():
              c5 = cc.permute(0, 2, 3, 1).reshape(B, 1, hh2, ww2, 2).contiguous()
              base = None
              line = f"altcorr {hh2}x{ww2} B={B} {fname:8s}:"
              for mode in (1, 2, 3, 4, 0):
                  ops.debug_set_altcorr(mode)
                  for _ in range(3):
                      o = ops.altcorr_forward(f1, f2, c5, r)
                  e0.record()
                  for _ in range(20):
                      o = ops.altcorr_forward(f1, f2, c5, r)
                  e1.record(); torch.cuda.synchronize()
                  us = 1e3 * e0.elapsed_time(e1) / 20
                  o = o[0] if isinstance(o, (list, tuple)) else o
                  if base is None:
                      base = o.clone()
                  err = float((o - base).abs().max() / base.abs().max())
                  line += f" mode {mode}: {us:7.1f} us ({B*Nn*100*256*2/us/1e6:5.1f} TF useful, diff {err:.1e}) |"
              ops.debug_set_altcorr(0)
              print(line)
              # bf16 feature maps (pfk_altcorr_forward_bf16): per-pixel kernel and the library's choice
              f1b, f2b = f1.bfloat16(), f2.bfloat16()
              line = f"altcorr {hh2}x{ww2} B={B} {fname:8s} bf16 maps:"
              for mode in (1, 0):
                  ops.debug_set_altcorr(mode)
                  for _ in range(3):
                      o = ops.altcorr_forward(f1b, f2b, c5, r)
                  e0.record()
                  for _ in range(20):
                      o = ops.altcorr_forward(f1b, f2b, c5, r)
                  e1.record(); torch.cuda.synchronize()
                  us = 1e3 * e0.elapsed_time(e1) / 20
                  line += f" mode {mode}: {us:7.1f} us |"
              ops.debug_set_altcorr(0)
              print(line)
