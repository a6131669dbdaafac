"""TEST INFRASTRUCTURE — generates tests/golden/*.pt by running the *reference's own code*
(/root/reference, imported unmodified through oracle/ref_loader.py) on seeded synthetic inputs.

    python -m oracle.make_golden            # needs /root/reference; run in the build container

The reference has no test that pins numbers on this path (test_forward only checks
"does not raise", test_accuracy is skipped and needs network checkpoints), so these vectors ARE the
pin: tests/test_oracle_golden.py holds oracle/raft_oracle.py to them, and the GPU tests hold the HIP
kernels to the oracle (and to these vectors directly).  Weights are not stored: they are regenerated
from `ptlflow_amd.synth` (seeded, construction-order independent); only inputs/outputs are saved.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import raft_oracle as O  # noqa: E402
from oracle import ref_loader  # noqa: E402
from ptlflow_amd.synth import synth_state_dict, synth_update_block_params  # noqa: E402
from ptlflow_amd.update import basic_spec, small_spec  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def coords_cases(B, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    c0 = O.coords_grid(B, h, w)
    cases = {
        "integer": c0,
        "fractional": c0 + torch.rand(B, 2, h, w, generator=g) * 16 - 8,
        "out_of_bounds": c0 + torch.randn(B, 2, h, w, generator=g) * 40,
    }
    bad = c0.clone()
    bad[:, 0, 0, 0] = float("nan")
    bad[:, 1, 0, 1] = float("inf")
    bad[:, 0, 1, 0] = -float("inf")
    bad[:, 0, 1, 1] = 3.0e9
    cases["nonfinite"] = bad
    return cases


def golden_corr():
    """CorrBlock of raft (avg-pool pyramid) and sea_raft (per-level GEMM) + lookups."""
    raft_corr = ref_loader.ref_module("ptlflow.models.raft.corr")
    sea_corr = ref_loader.ref_module("ptlflow.models.sea_raft.corr")
    out = {}
    for tag, (h, w, D, L, r) in {"raft_10x16": (10, 16, 32, 4, 4), "small_8x16_degenerate": (8, 16, 32, 4, 3),
                                 "ccmr_9x13_L2": (9, 13, 32, 2, 4)}.items():
        g = torch.Generator().manual_seed(len(tag) * 17)
        f1 = torch.randn(1, D, h, w, generator=g)
        f2 = torch.randn(1, D, h, w, generator=g)
        cb = raft_corr.CorrBlock(f1, f2, num_levels=L, radius=r)
        item = {"fmap1": f1, "fmap2": f2, "levels": L, "radius": r,
                "pyramid": [p.clone() for p in cb.corr_pyramid], "lookups": {}}
        for name, c in coords_cases(1, h, w, 5).items():
            item["lookups"][name] = {"coords": c, "out": cb(c).clone()}
        out[tag] = item
    # SEA-RAFT pyramid (sea_raft/corr.py:71-117)
    g = torch.Generator().manual_seed(99)
    f1 = torch.randn(1, 32, 16, 24, generator=g)
    f2 = torch.randn(1, 32, 16, 24, generator=g)
    cb = sea_corr.CorrBlock(f1, f2, num_levels=4, radius=4)
    c = O.coords_grid(1, 16, 24) + torch.rand(1, 2, 16, 24, generator=g) * 6 - 3
    out["sea_16x24"] = {"fmap1": f1, "fmap2": f2, "levels": 4, "radius": 4,
                        "pyramid": [p.clone() for p in cb.corr_pyramid],
                        "lookups": {"fractional": {"coords": c, "out": cb(c).clone()}}}
    torch.save(out, os.path.join(OUT, "corr_lookup.pt"))


def golden_update():
    upd = ref_loader.ref_module("ptlflow.models.raft.update")
    gma_upd = ref_loader.ref_module("ptlflow.models.gma.update")
    out = {}
    for tag, spec, cls, kw in (("basic", basic_spec(), upd.BasicUpdateBlock, dict(corr_levels=4, corr_radius=4, hidden_dim=128)),
                               ("small", small_spec(), upd.SmallUpdateBlock, dict(corr_levels=4, corr_radius=3, hidden_dim=96))):
        P = synth_update_block_params(spec, seed=21)
        m = cls(**kw).eval()
        m.load_state_dict(P, strict=True)
        g = torch.Generator().manual_seed(31)
        B, H, W = 1, 8, 12
        net = torch.tanh(torch.randn(B, spec.hidden, H, W, generator=g))
        inp = torch.relu(torch.randn(B, spec.context, H, W, generator=g))
        corr = torch.randn(B, spec.corr_channels, H, W, generator=g)
        flow = torch.randn(B, 2, H, W, generator=g) * 3
        with torch.no_grad():
            n, mk, d = m(net, inp, corr, flow)
        out[tag] = {"seed": 21, "net": net, "inp": inp, "corr": corr, "flow": flow,
                    "net_out": n.clone(), "mask_out": None if mk is None else mk.clone(), "delta_out": d.clone()}
    # SepConvGRU with the GMA / CCMR input width (C_in = 128 + 384 = 512; gma/update.py:135-137)
    gru = gma_upd.SepConvGRU(hidden_dim=128, input_dim=384).eval()
    shapes = {k: tuple(v.shape) for k, v in gru.state_dict().items()}
    Pg = synth_state_dict(shapes, seed=22)
    gru.load_state_dict(Pg, strict=True)
    g = torch.Generator().manual_seed(32)
    h = torch.tanh(torch.randn(1, 128, 8, 12, generator=g))
    x = torch.randn(1, 384, 8, 12, generator=g)
    with torch.no_grad():
        ho = gru(h, x)
    out["sepconvgru_512"] = {"seed": 22, "shapes": shapes, "h": h, "x": x, "h_out": ho.clone()}
    torch.save(out, os.path.join(OUT, "update_block.pt"))


def golden_forward():
    out = {}
    for tag, small, H, W, iters in (("raft_128x160_it6", False, 128, 160, 6), ("raft_small_128x128_it6", True, 128, 128, 6)):
        ref = ref_loader.build_raft(small=small, iters=iters)
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.startswith("train_metrics")}
        P = synth_state_dict(shapes, seed=41)
        missing = ref.load_state_dict(P, strict=False)
        assert not missing.unexpected_keys and all(k.startswith("train_metrics") for k in missing.missing_keys)
        x = O.smooth_pair(1, H, W, seed=43)
        with torch.no_grad():
            o = ref({"images": x.clone()})
        out[tag] = {"seed": 41, "shapes": shapes, "small": small, "iters": iters, "images": x,
                    "flows": o["flows"].clone(), "flow_small": o["flow_small"].clone()}
    torch.save(out, os.path.join(OUT, "raft_forward.pt"))


def golden_gma():
    """GMA (gma/gma.py) whole forward + one GMAUpdateBlock step with a real attention map."""
    g = ref_loader.ref_module("ptlflow.models.gma.gma")
    ref = g.GMA(iters=4).eval()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()
              if not k.startswith("train_metrics") and v.is_floating_point()}
    P = synth_state_dict(shapes, seed=51)
    missing = ref.load_state_dict(P, strict=False)
    assert not missing.unexpected_keys
    x = O.smooth_pair(1, 128, 160, seed=53)
    with torch.no_grad():
        o = ref({"images": x.clone()})
    out = {"forward": {"seed": 51, "shapes": shapes, "iters": 4, "images": x, "flows": o["flows"].clone(),
                       "flow_small": o["flow_small"].clone()}}
    gen = torch.Generator().manual_seed(54)
    B, H, W = 1, 8, 12
    net = torch.tanh(torch.randn(B, 128, H, W, generator=gen))
    inp = torch.relu(torch.randn(B, 128, H, W, generator=gen))
    corr = torch.randn(B, 324, H, W, generator=gen)
    flow = torch.randn(B, 2, H, W, generator=gen) * 3
    with torch.no_grad():
        attn = ref.att(inp)
        n, mk, d = ref.update_block(net, inp, corr, flow, attn)
    out["update"] = {"net": net, "inp": inp, "corr": corr, "flow": flow, "attention": attn.clone(),
                     "net_out": n.clone(), "mask_out": mk.clone(), "delta_out": d.clone()}
    torch.save(out, os.path.join(OUT, "gma.pt"))


def golden_warm_start():
    """forward_interpolate (utils/external/raft.py:155-185, scipy griddata on the host) on random and structured flows,
    and a warm-started RAFT forward (`prev_preds`, raft.py:162-167)."""
    ext = ref_loader.ref_module("ptlflow.utils.external.raft")
    gen = torch.Generator().manual_seed(61)
    cases = {}
    for tag, h, w, scale in (("55x128_s6", 55, 128, 6.0), ("20x33_s3", 20, 33, 3.0), ("16x24_s40_mostly_outside", 16, 24, 40.0)):
        f = torch.randn(2, h, w, generator=gen) * scale
        cases[tag] = {"flow": f, "out": ext.forward_interpolate(f)}
    f = torch.zeros(2, 12, 20)
    f[0] = 2.25
    f[1] = -1.5
    cases["12x20_constant"] = {"flow": f, "out": ext.forward_interpolate(f)}
    ref = ref_loader.build_raft(small=False, iters=4)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.startswith("train_metrics")}
    ref.load_state_dict(synth_state_dict(shapes, seed=63), strict=False)
    x = O.smooth_pair(1, 128, 160, seed=65)
    with torch.no_grad():
        first = ref({"images": x.clone()})
        second = ref({"images": x.clone(), "prev_preds": {"flow_small": first["flow_small"].clone()}})
    out = {"interp": cases,
           "forward": {"seed": 63, "shapes": shapes, "iters": 4, "images": x, "prev_flow_small": first["flow_small"].clone(),
                       "flows": second["flows"].clone(), "flow_small": second["flow_small"].clone()}}
    torch.save(out, os.path.join(OUT, "warm_start.pt"))


def golden_sea_raft_model():
    """The correlation path of the REAL SEA-RAFT model in one forward (sea_raft/sea_raft.py:209-224): the two 1/8-resolution
    feature maps its `get_corr_block` call receives (D = 256, ResNet-FPN features of a seeded random-init model on a smooth
    frame pair) and, per refinement iteration, the coordinates handed to `corr_fn` and what it returned.  The weights are not
    needed to replay this: the fixture is self-contained."""
    mod = ref_loader.ref_module("ptlflow.models.sea_raft.sea_raft")
    torch.manual_seed(1234)
    model = mod.SEARAFT(block_dims=[64, 128, 256], iters=4).eval()
    rec = {"calls": []}
    orig = mod.get_corr_block

    def spy(fmap1, fmap2, radius, num_levels, alternate_corr=False):
        cb = orig(fmap1=fmap1, fmap2=fmap2, radius=radius, num_levels=num_levels, alternate_corr=alternate_corr)
        rec.update(fmap1=fmap1.detach().clone(), fmap2=fmap2.detach().clone(), radius=radius, levels=num_levels,
                   pyramid_shapes=[tuple(p.shape) for p in cb.corr_pyramid])

        def call(coords):
            out = cb(coords)
            rec["calls"].append({"coords": coords.detach().clone(), "out": out.detach().clone()})
            return out

        return call

    mod.get_corr_block = spy
    try:
        x = O.smooth_pair(1, 128, 192, seed=47)
        with torch.no_grad():
            model({"images": x})
    finally:
        mod.get_corr_block = orig
    assert len(rec["calls"]) == 4 and rec["fmap1"].shape == (1, 256, 16, 24)
    torch.save(rec, os.path.join(OUT, "sea_raft_model.pt"))


def golden_alt_corr():
    """The on-demand correlation path as the reference computes it WITHOUT its CUDA extension: `IterativeCorrBlock`
    (ptlflow/utils/correlation.py:539-615, what raft/corr.py:111-113 builds for `alternate_corr=True` when `alt_cuda_corr` is
    missing) on a 64x136 grid — 272 patches of 8x4 pixels, enough for libpfk's window-sharing kernel to be the one that runs —
    with a smooth flow field plus 0.6 px of per-pixel noise (+0.013: away from exact integers, where grid_sample's index round trip
    and a raw floor() disagree).  Output rows 0, 4, 8, ... are stored (1.4 MB instead of 5.6)."""
    corr_mod = ref_loader.ref_module("ptlflow.models.raft.corr")
    g = torch.Generator().manual_seed(77)
    B, C, H, W, L, r = 1, 32, 64, 136, 2, 4
    f1, f2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    smooth = torch.nn.functional.interpolate(torch.randn(B, 2, H // 8 + 2, W // 8 + 2, generator=g) * 2.5, size=(H, W), mode="bicubic",
                                             align_corners=True)
    coords = O.coords_grid(B, H, W) + smooth + torch.randn(B, 2, H, W, generator=g) * 0.6 + 0.013
    with torch.no_grad():
        out = corr_mod.IterativeCorrBlock(fmap1=f1, fmap2=f2, radius=r, num_levels=L)(coords)
    assert tuple(out.shape) == (B, L * (2 * r + 1) ** 2, H, W)
    torch.save({"fmap1": f1, "fmap2": f2, "coords": coords, "levels": L, "radius": r, "row_step": 4,
                "out_rows": out[:, :, ::4].clone()}, os.path.join(OUT, "alt_corr.pt"))


def golden_alt_corr_bf16():
    """The same path for bf16 callers: `IterativeCorrBlock` on bf16 feature maps under `torch.autocast("cpu", bfloat16)` (the
    reference has no bf16 mode of its own; autocast is how BASELINE config 3's precision reaches it) on golden_alt_corr's
    inputs.  Stored: rows 0, 4, 8, ... of the autocast output and the largest difference of that output from the fp32 one —
    the reference's own bf16 gap, which `pfk_altcorr_forward_bf16` (exact products of the bf16 values) must stay inside."""
    corr_mod = ref_loader.ref_module("ptlflow.models.raft.corr")
    gold = torch.load(os.path.join(OUT, "alt_corr.pt"))
    f1, f2, coords, L, r = gold["fmap1"], gold["fmap2"], gold["coords"], gold["levels"], gold["radius"]
    with torch.no_grad():
        ref32 = corr_mod.IterativeCorrBlock(fmap1=f1, fmap2=f2, radius=r, num_levels=L)(coords)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = corr_mod.IterativeCorrBlock(fmap1=f1.bfloat16(), fmap2=f2.bfloat16(), radius=r, num_levels=L)(coords).float()
    torch.save({"row_step": 4, "out_rows_autocast": out[:, :, ::4].clone(), "autocast_gap_max": float((out - ref32).abs().max()),
                "autocast_gap_mean": float((out - ref32).abs().mean())}, os.path.join(OUT, "alt_corr_bf16.pt"))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    golden_corr()
    golden_update()
    golden_forward()
    golden_gma()
    golden_warm_start()
    golden_sea_raft_model()
    golden_alt_corr()
    golden_alt_corr_bf16()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
