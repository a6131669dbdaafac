"""BASELINE config 3 (gma + the shared CorrBlock / GRU path in bf16): the parity gate SURVEY.md §8d prescribes.

BASELINE.json asks for bf16; the reference itself has no bf16 switch — its only reduced-precision mode is `model.half()`
(fp16: validate.py:243-244, model_benchmark.py:317-319) — so the bf16 oracle is the reference's forward run under
`torch.autocast("cpu", bfloat16)`, the standard way to put that forward in bf16.
Its own distance from the fp32 forward — the "autocast-CPU gap" — is measured here, on the same seeded input, by running the
CPU oracle under `torch.autocast("cpu", bfloat16)` (bit-identical to the live reference under autocast:
tests/test_oracle_vs_reference.py::test_oracle_under_autocast_is_the_reference_under_autocast).  Gate:

    EPE(GPU bf16 mode, CPU fp32)  <=  2 x EPE(CPU autocast-bf16, CPU fp32)          (mean end-point error)

GPU bf16 mode = `conv_precision="bf16"`: bf16 operands on the matrix cores for every convolution (encoders + update block)
and for the correlation volume (`pfk_corr_volume_bf16`, bf16 pyramid, `pfk_corr_lookup_bf16`); accumulation, the recurrent
state, coordinates and the lookup arithmetic stay fp32."""
import pytest
import torch

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


def _gate(kind, H, W, iters, gpu, seed):
    from ptlflow_amd.raft import GMA, RAFT
    make = (lambda **kw: GMA(iters=iters, **kw)) if kind == "gma" else (lambda **kw: RAFT(iters=iters, **kw))
    base = make().load_synthetic(seed).eval()
    P = {k: v.clone() for k, v in base.state_dict().items()}
    x = O.smooth_pair(1, H, W, seed)
    from _cpu_cache import cpu_forward      # the fp32 CPU forward of the headline configuration is shared with test_gpu_model.py
    key = ("synthetic", seed, "smooth", seed)
    ref32 = cpu_forward(kind, P, x, iters, key=key)["flows"][:, 0]
    ref_ac = cpu_forward(kind, P, x, iters, autocast=True, key=key)["flows"][:, 0].float()
    gap, gap_max = O.epe(ref_ac, ref32)
    m = make(conv_precision="bf16").eval()
    m.load_state_dict(P)
    out = m.to(gpu)({"images": x.to(gpu)})["flows"][:, 0].float().cpu()
    mean, mx = O.epe(out, ref32)
    print(f"{kind} {H}x{W} {iters} it: autocast-CPU gap mean {gap:.3e} max {gap_max:.3e} | GPU bf16 mean {mean:.3e} max {mx:.3e}")
    assert gap > 0
    assert mean <= 2 * gap, f"GPU bf16 EPE {mean:.3e} exceeds 2x the autocast-CPU gap {gap:.3e}"
    return mean, gap


def test_raft_bf16_gate_headline(gpu):
    """raft, 436x1024, 32 iterations."""
    _gate("raft", 436, 1024, 32, gpu, 1234)


def test_gma_bf16_gate_headline(gpu):
    """gma, 436x1024, 32 iterations."""
    _gate("gma", 436, 1024, 32, gpu, 1234)


def test_bf16_pyramid_and_lookup(gpu):
    """`pfk_corr_volume_bf16` + `pfk_corr_pool2x2_bf16` + `pfk_corr_lookup_bf16` against the oracle's pyramid built from bf16
    operands the way autocast builds it (bmm in bf16, / sqrt(D) in bf16, avg_pool2d in bf16): volume within one bf16 ulp,
    lookup of the SAME bf16 pyramid bit-exact (its arithmetic is fp32)."""
    from ptlflow_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(6)
    B, D, h, w = 2, 256, 24, 40
    f1 = torch.randn(B, D, h, w, generator=g).bfloat16().float()     # bf16-representable, as an autocast encoder emits
    f2 = torch.randn(B, D, h, w, generator=g).bfloat16().float()
    pyr = O.correlation_pyramid(f1.bfloat16(), f2.bfloat16(), 4)
    cb = CorrBlock(f1.to(gpu), f2.to(gpu), 4, 4, volume_dtype=torch.bfloat16)
    assert all(p.dtype == torch.bfloat16 for p in cb.corr_pyramid)
    import torch.nn.functional as F
    a32, b32 = cb.corr_pyramid[0].float().cpu().reshape(pyr[0].shape), pyr[0].float()
    # level 0: fp32-accumulated, once-rounded (GPU) vs bf16 matmul then bf16 division (oracle): at most one bf16 ulp apart
    assert ((a32 - b32).abs() <= 2.0 ** -7 * b32.abs().clamp_min(2.0 ** -6)).all()
    assert ((a32 - b32) != 0).float().mean().item() < 1e-2          # ... and that only on rounding ties of the accumulation order
    # levels 1..: the kernel's pool of ITS OWN previous level is what torch's bf16 avg_pool2d gives (fp32 window sum, one rounding)
    for prev, cur in zip(cb.corr_pyramid[:-1], cb.corr_pyramid[1:]):
        assert torch.equal(F.avg_pool2d(prev.cpu().unsqueeze(1), 2, stride=2).squeeze(1), cur.cpu())
    c = O.coords_grid(B, h, w) + torch.rand(B, 2, h, w, generator=g) * 8 - 4
    got = cb(c.to(gpu)).cpu()
    ref = O.lookup([p.float().cpu() for p in cb.corr_pyramid], c, 4)      # same pyramid, oracle lookup
    assert got.dtype == torch.float32 and torch.equal(got, ref)
