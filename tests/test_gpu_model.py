"""End-to-end parity: ptlflow_amd.RAFT on the MI355X vs the CPU oracle on identical seeded weights/inputs.

Gate (BASELINE.json north_star): end-point error of `flows` <= 1e-3 px (mean) in fp32; we also bound the
max.  The oracle itself differs from an fp64 run of the same model by ~1e-5 mean / 5e-5 max after 32
iterations, so the gate is meaningful."""
import pytest
import torch

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


def _run_pair(small, H, W, iters, B=1, seed=1234, upsample_every_iter=True, smooth=True):
    from ptlflow_amd.raft import RAFT
    from ptlflow_amd.synth import rand_pair
    model = RAFT(small=small, iters=iters, upsample_every_iter=upsample_every_iter).load_synthetic(seed).eval()
    P = {k: v.clone() for k, v in model.state_dict().items()}
    x = O.smooth_pair(B, H, W, seed) if smooth else rand_pair(B, H, W, seed)
    from _cpu_cache import cpu_forward
    ref = cpu_forward("raft", P, x, iters, small=small, key=("synthetic", seed, "smooth" if smooth else "rand", seed))
    model = model.cuda()
    out = model({"images": x.cuda()})
    torch.cuda.synchronize()
    return out, ref


@pytest.mark.parametrize("small,H,W,iters", [(False, 128, 192, 4), (False, 184, 320, 12), (True, 128, 256, 12)])
def test_raft_small_shapes(gpu, small, H, W, iters):
    out, ref = _run_pair(small, H, W, iters)
    assert out["flows"].shape == ref["flows"].shape
    mean, mx = O.epe(out["flows"][:, 0].cpu(), ref["flows"][:, 0])
    mean_s, mx_s = O.epe(out["flow_small"].cpu(), ref["flow_small"])
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"
    assert mean_s <= 1e-3 and mx_s <= 1e-2, f"flow_small EPE mean {mean_s:.2e} max {mx_s:.2e}"


def test_raft_headline_config(gpu):
    """BASELINE.json configs[1]: raft, 436x1024 (Sintel), 32 iterations, fp32."""
    out, ref = _run_pair(False, 436, 1024, 32)
    assert tuple(out["flows"].shape) == (1, 1, 2, 436, 1024)
    mean, mx = O.epe(out["flows"][:, 0].cpu(), ref["flows"][:, 0])
    print(f"headline EPE mean {mean:.3e} max {mx:.3e}")
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


def test_skip_dead_upsample_is_output_identical(gpu):
    """Skipping mask head + upsampling on non-final eval iterations (dead work in raft.py:180-187) must not
    change a single output bit."""
    a, _ = _run_pair(False, 128, 192, 5, upsample_every_iter=True)
    b, _ = _run_pair(False, 128, 192, 5, upsample_every_iter=False)
    assert torch.equal(a["flows"], b["flows"]) and torch.equal(a["flow_small"], b["flow_small"])


def test_degenerate_64x128_nan_pattern(gpu):
    """BASELINE.json configs[0]: raft_small at 64x128 is all-NaN in the reference (1-pixel pyramid level,
    SURVEY finding 4); the accelerated path must reproduce the NaN pattern, not hide it."""
    out, ref = _run_pair(True, 64, 128, 3, smooth=False)
    assert torch.equal(torch.isnan(out["flows"].cpu()), torch.isnan(ref["flows"]))
    assert bool(torch.isnan(ref["flows"]).all())


def test_batch_independence(gpu):
    """Frame pairs are independent units: a batch of 2 equals two batches of 1 (the convolution
    kernels pick their schedule — tile grid vs stream-K split — by grid size, which changes a tile's summation order, hence a
    tolerance, not bit-equality)."""
    from ptlflow_amd.raft import RAFT
    m = RAFT(iters=3).load_synthetic(7).eval().cuda()
    x = O.smooth_pair(2, 128, 160, 3).cuda()
    both = m({"images": x})["flows"]
    one = torch.cat([m({"images": x[i:i + 1]})["flows"] for i in range(2)], 0)
    mean, mx = O.epe(both[:, 0].cpu(), one[:, 0].cpu())
    assert mean <= 1e-4 and mx <= 1e-3, f"EPE mean {mean:.2e} max {mx:.2e}"


def test_gma_forward(gpu):
    """Second model family on the same kernels: GMA, 6 iterations, vs the CPU oracle."""
    from ptlflow_amd.raft import GMA
    model = GMA(iters=6).load_synthetic(77).eval()
    P = {k: v.clone() for k, v in model.state_dict().items()}
    assert abs(float(P["update_block.aggregator.gamma"])) > 0.1      # the aggregate branch is really exercised
    x = O.smooth_pair(2, 184, 248, seed=9)
    ref = O.gma_forward(P, x, iters=6)
    out = model.cuda()({"images": x.cuda()})
    mean, mx = O.epe(out["flows"][:, 0].cpu(), ref["flows"][:, 0])
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


def test_gma_headline_config(gpu):
    """BASELINE.json configs[2], fp32 side: gma at 436x1024, 32 iterations (attention map 7040 x 7040, aggregation GEMM and the
    512-input SepConvGRU on libpfk) vs the CPU oracle — the shape the `config3.gma_fp32` bench leg times."""
    from ptlflow_amd.raft import GMA
    model = GMA(iters=32).load_synthetic(1234).eval()
    P = {k: v.clone() for k, v in model.state_dict().items()}
    x = O.smooth_pair(1, 436, 1024, seed=1234)
    from _cpu_cache import cpu_forward
    ref = cpu_forward("gma", P, x, 32, key=("synthetic", 1234, "smooth", 1234))
    out = model.cuda()({"images": x.cuda()})
    assert tuple(out["flows"].shape) == (1, 1, 2, 436, 1024)
    mean, mx = O.epe(out["flows"][:, 0].cpu(), ref["flows"][:, 0])
    print(f"gma headline EPE mean {mean:.3e} max {mx:.3e}")
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


def test_gma_odd_grid(gpu):
    """h*w not a multiple of 4 / 32: the attention operand and V^T get padded."""
    from ptlflow_amd.raft import GMA
    model = GMA(iters=3).load_synthetic(78).eval()
    P = {k: v.clone() for k, v in model.state_dict().items()}
    x = O.smooth_pair(1, 136, 168, seed=10)          # 17 x 21 = 357 pixels
    ref = O.gma_forward(P, x, iters=3)
    out = model.cuda()({"images": x.cuda()})
    mean, mx = O.epe(out["flows"][:, 0].cpu(), ref["flows"][:, 0])
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


@pytest.mark.parametrize("H,W,B,iters", [(375, 1242, 2, 4), (368, 496, 2, 4)])
def test_other_baseline_shapes(gpu, H, W, B, iters):
    """BASELINE.json configs[3] (KITTI 1242x375 -> 47x156 grid, N = 7332: not a multiple of 32/64) and configs[4]'s
    FlyingChairs crop (368x496 -> 46x62), a few iterations, batch > 1."""
    out, ref = _run_pair(False, H, W, iters, B=B)
    assert out["flows"].shape == ref["flows"].shape == (B, 1, 2, H, W)
    mean, mx = O.epe(out["flows"][:, 0].cpu(), ref["flows"][:, 0])
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


def test_kitti_config_batch8_32_iterations(gpu):
    """BASELINE.json configs[3] as it is quoted — raft on KITTI 375x1242 pairs (47x156 grid, N = 7332), 32 iterations, 8 pairs per GPU:
    the batch-8 forward against the CPU oracle on its last pair (the oracle takes ~10 s per pair), and pair by pair
    against single-pair GPU forwards (a pair's result must not depend on what shares its batch: EPE <= 1e-4, a tenth of the gate —
    batch 1 and batch 8 pick different tile schedules, so rounding may differ)."""
    from ptlflow_amd.raft import RAFT
    H, W, B = 375, 1242, 8
    model = RAFT(iters=32).load_synthetic(1234).eval()
    P = {k: v.clone() for k, v in model.state_dict().items()}
    x = O.smooth_pair(B, H, W, 1234)
    model = model.cuda()
    out = model({"images": x.cuda()})["flows"][:, 0].float().cpu()
    assert tuple(out.shape) == (B, 2, H, W)
    for b in (B - 1,):
        ref = O.raft_forward(P, x[b:b + 1], iters=32)["flows"][:, 0]
        mean, mx = O.epe(out[b:b + 1], ref)
        assert mean <= 1e-3 and mx <= 1e-2, f"pair {b}: EPE mean {mean:.2e} max {mx:.2e}"
    for b in (0, 4):
        one = model({"images": x[b:b + 1].cuda()})["flows"][:, 0].float().cpu()
        mean, mx = O.epe(out[b:b + 1], one)
        assert mean <= 1e-4, f"pair {b} inside the batch vs alone: EPE mean {mean:.2e} max {mx:.2e}"
    assert O.epe(out[0:1], out[1:2])[0] > 1e-3      # (the pairs differ: smooth_pair seeds each pair of a batch differently)


def test_warm_start_kernel_bit_exact(gpu):
    """pfk_forward_interpolate_f32 vs the reference's scipy result (golden) and vs the oracle on fresh random flows."""
    import os
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "warm_start.pt"), weights_only=False)
    for tag, c in g["interp"].items():
        f = c["flow"][None].contiguous().cuda()
        out = torch.empty_like(f)
        torch.ops.pfk.forward_interpolate(f, out)
        assert torch.equal(out[0].cpu(), c["out"]), tag
    gen = torch.Generator().manual_seed(5)
    f = torch.randn(3, 2, 23, 41, generator=gen) * 5
    out = torch.empty_like(f).cuda()
    torch.ops.pfk.forward_interpolate(f.cuda(), out)
    assert torch.equal(out.cpu(), O.forward_interpolate_batch(f))


def test_warm_started_forward(gpu):
    """`prev_preds` warm start (raft.py:162-167) through the device kernel vs the reference's own output (golden)."""
    import os
    from ptlflow_amd.raft import RAFT
    from ptlflow_amd.synth import synth_state_dict
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "warm_start.pt"), weights_only=False)["forward"]
    model = RAFT(iters=g["iters"]).eval()
    model.load_state_dict(synth_state_dict(g["shapes"], seed=g["seed"]))
    out = model.cuda()({"images": g["images"].cuda(), "prev_preds": {"flow_small": g["prev_flow_small"].cuda()}})
    mean, mx = O.epe(out["flows"][:, 0].cpu(), g["flows"][:, 0])
    assert mean <= 1e-3 and mx <= 1e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


def test_alternate_corr_forward(gpu):
    """RAFT(alternate_corr=True): on-demand correlation instead of the materialised volume.  The on-demand kernel follows the
    reference CUDA kernel's raw floor() taps, the volume path grid_sample's round trip, so the two forwards agree closely but
    not bit for bit (SURVEY finding 3); both stay inside the EPE gate against each other."""
    from ptlflow_amd.raft import RAFT
    x = O.smooth_pair(1, 128, 192, seed=4).cuda()
    a = RAFT(iters=4).load_synthetic(9).eval().cuda()
    b = RAFT(iters=4, alternate_corr=True).load_synthetic(9).eval().cuda()
    fa, fb = a({"images": x})["flows"], b({"images": x})["flows"]
    mean, mx = O.epe(fa[:, 0].cpu(), fb[:, 0].cpu())
    assert mean <= 5e-3 and mx <= 5e-2, f"EPE mean {mean:.2e} max {mx:.2e}"


@pytest.mark.parametrize("small", [False, True])
def test_graph_replay_matches_eager(gpu, small):
    """use_graph=True: the iteration loop recorded into a hipGraph must reproduce the eager forward bit for bit, for new inputs
    of the same shape, after a shape change in between, and with a warm start (raft and raft_small: the latter's upflow8 writes
    into a fixed buffer too)."""
    from ptlflow_amd.raft import RAFT
    eager = RAFT(iters=5, small=small, use_graph=False).load_synthetic(11).eval().cuda()
    graph = RAFT(iters=5, small=small, use_graph=True).load_synthetic(11).eval().cuda()      # forked branches inside the graph
    xs = [O.smooth_pair(1, 128, 192, seed=s).cuda() for s in (1, 2, 3)]
    other = O.smooth_pair(1, 136, 160, seed=7).cuda()
    prev = None
    for i, x in enumerate(xs + [other] + xs[:2]):
        inp = {"images": x}
        if i == 2 and prev is not None:
            inp["prev_preds"] = {"flow_small": prev}
        a, b = eager(dict(inp)), graph(dict(inp))
        assert torch.equal(a["flows"], b["flows"]) and torch.equal(a["flow_small"], b["flow_small"]), f"forward {i}"
        prev = a["flow_small"] if x.shape == xs[0].shape else None
    assert len(graph._graphs) == 2


@pytest.mark.parametrize("small,B,H,W,every", [(False, 1, 436, 1024, True), (False, 2, 184, 320, False), (True, 1, 184, 320, True)])
def test_forked_branches_are_bit_identical(gpu, small, B, H, W, every):
    """`fork_branches=True` (what the captured graph uses): the motion encoder's flow branch next to lookup -> convc1 -> convc2,
    mask conv2 + upsampling next to the coordinate update and the following iteration, cnet next to fnet — the same launches on
    the same operands with per-branch stream-K workspaces, so every output bit must equal the serial schedule's; repeated
    forwards on fresh inputs (a missing event shows up as a stale buffer), eagerly and inside the captured graph."""
    from ptlflow_amd.raft import RAFT
    serial = RAFT(iters=6, small=small, upsample_every_iter=every, use_graph=False, fork_branches=False).load_synthetic(5).eval().cuda()
    forked = RAFT(iters=6, small=small, upsample_every_iter=every, use_graph=False, fork_branches=True).load_synthetic(5).eval().cuda()
    auto = RAFT(iters=6, small=small, upsample_every_iter=every, use_graph=True).load_synthetic(5).eval().cuda()
    for seed in (1, 2, 1, 3):
        x = O.smooth_pair(B, H, W, seed=seed).cuda()
        a, b, c = serial({"images": x}), forked({"images": x}), auto({"images": x})
        torch.cuda.synchronize()
        assert torch.equal(a["flows"], b["flows"]) and torch.equal(a["flow_small"], b["flow_small"]), f"forked, seed {seed}"
        assert torch.equal(a["flows"], c["flows"]) and torch.equal(a["flow_small"], c["flow_small"]), f"graph, seed {seed}"
    assert len(auto._graphs) == 1


@pytest.mark.parametrize("kind,B,H,W,every", [("raft", 1, 436, 1024, True), ("raft", 2, 184, 320, False), ("raft_small", 1, 184, 320, True),
                                             ("gma", 1, 184, 320, True)])
def test_grouped_launches_are_bit_identical(gpu, kind, B, H, W, every):
    """`group_launches` (the small-batch default): convc1 | convf2 | the previous iteration's mask conv2 in ONE grid
    (`pfk_conv2d_group_f32`), the previous iteration's upsampling behind it — the same tiles with the same K order on the same
    operands as the one-launch-per-convolution schedule, so the outputs must not differ by a bit; repeated forwards on fresh inputs
    (a wrong ordering against the flow slice / `fm` would show as a stale operand), eagerly and in the captured graph."""
    from ptlflow_amd.raft import GMA, RAFT
    if kind == "gma":
        make = lambda **kw: GMA(iters=6, upsample_every_iter=every, **kw)                           # noqa: E731
    else:
        make = lambda **kw: RAFT(iters=6, small=kind == "raft_small", upsample_every_iter=every, **kw)  # noqa: E731
    single, grouped = make().load_synthetic(5).eval().cuda(), make().load_synthetic(5).eval().cuda()
    single.group_launches, grouped.group_launches = False, True
    models = [single, grouped]
    if kind != "gma":
        graph = make(use_graph=True).load_synthetic(5).eval().cuda()
        graph.group_launches = True
        graph.fork_branches = False
        models.append(graph)
    for seed in (1, 2, 1, 3):
        x = O.smooth_pair(B, H, W, seed=seed).cuda()
        outs = [m({"images": x}) for m in models]
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert torch.equal(outs[0]["flows"], o["flows"]) and torch.equal(outs[0]["flow_small"], o["flow_small"]), f"seed {seed}"


@pytest.mark.parametrize("kind", ["raft", "gma"])
def test_side_stream_mask_head_is_bit_identical(gpu, kind):
    """overlap_mask_head=True runs mask conv2 + convex upsampling of iteration i on a second stream next to iteration i+1: same
    kernels on the same operands, so the flows must not change by a bit — over repeated forwards (buffer re-use across forwards and
    iterations is where a missing event would show) and with the per-iteration upsampling both on and off."""
    from ptlflow_amd.raft import GMA, RAFT
    make = (lambda **kw: GMA(iters=6, **kw)) if kind == "gma" else (lambda **kw: RAFT(iters=6, **kw))
    for every in (True, False):
        a = make(upsample_every_iter=every).load_synthetic(5).eval().cuda()
        b = make(upsample_every_iter=every).load_synthetic(5).eval().cuda()
        a.overlap_mask_head, b.overlap_mask_head = False, True
        for seed in (1, 2, 1, 3):
            x = O.smooth_pair(8, 480, 640, seed=seed).cuda()      # 8 x 60 x 80 = 38 400 pixels: above the side-stream threshold
            fa, fb = a({"images": x}), b({"images": x})
            assert torch.equal(fa["flows"], fb["flows"]) and torch.equal(fa["flow_small"], fb["flow_small"]), (every, seed)


@pytest.mark.parametrize("kind,B,H,W", [("raft", 8, 480, 640), ("raft", 1, 184, 320), ("gma", 2, 200, 328)])
def test_fused_mask_upsample_is_bit_identical(gpu, kind, B, H, W):
    """`fuse_mask_upsample`: mask conv2 + softmax + convex upsampling as ONE kernel that keeps the nine logits of a sub-pixel in
    registers (pfk_mask_upsample_f32) against the two separate launches through a [M, 576] mask in HBM — same K order in the
    convolution, the same epilogue / upsampling arithmetic operation for operation, so not a bit of `flows` may change; on the main
    stream and on the side stream, with per-iteration upsampling on and off, over repeated forwards."""
    from ptlflow_amd.raft import GMA, RAFT
    make = (lambda **kw: GMA(iters=5, **kw)) if kind == "gma" else (lambda **kw: RAFT(iters=5, **kw))
    for every in (True, False):
        for overlap in (True, False):
            a = make(upsample_every_iter=every).load_synthetic(5).eval().cuda()
            b = make(upsample_every_iter=every).load_synthetic(5).eval().cuda()
            a.fuse_mask_upsample, b.fuse_mask_upsample = False, True
            a.overlap_mask_head = b.overlap_mask_head = overlap
            assert b.engine(torch.device("cuda", 0)).can_fuse_mask_upsample
            for seed in (1, 2, 1):
                x = O.smooth_pair(B, H, W, seed=seed).cuda()
                fa, fb = a({"images": x}), b({"images": x})
                assert torch.equal(fa["flows"], fb["flows"]) and torch.equal(fa["flow_small"], fb["flow_small"]), (every, overlap, seed)


@pytest.mark.parametrize("kind,H,W,iters", [("raft", 436, 1024, 16), ("raft_small", 184, 320, 12), ("gma", 184, 320, 12)])
def test_hoisted_context_term_equals_the_single_chain_form(gpu, kind, H, W, iters):
    """The loop-invariant hoist (UpdateEngine: conv over cat([h, inp, m]) = conv over [h, m] + (conv over inp + bias), the second
    term once per forward) is the SAME sum in another association: against the single-chain launches (`hoist_context=False`)
    the flow may differ by fp32 rounding only — held here to 1e-4 px mean, a tenth of the north-star gate — and both stay
    inside the EPE gate against the CPU oracle.  One update-block step is also compared state by state (net, delta)."""
    from ptlflow_amd.raft import GMA, RAFT
    small = kind == "raft_small"
    outs = {}
    for hoist in (True, False):
        model = (GMA(iters=iters) if kind == "gma" else RAFT(small=small, iters=iters, hoist_context=hoist))
        if kind == "gma":
            model.hoist_context = hoist
        model = model.load_synthetic(1234).eval()
        P = {k: v.clone() for k, v in model.state_dict().items()}
        x = O.smooth_pair(1, H, W, 1234)
        model = model.cuda()
        outs[hoist] = model({"images": x.cuda()})
        assert model.engine(torch.device("cuda", 0)).hoist_context is hoist
    ref = O.gma_forward(P, x, iters=iters) if kind == "gma" else O.raft_forward(P, x, iters=iters, small=small)
    d_mean, d_max = O.epe(outs[True]["flows"][:, 0].cpu(), outs[False]["flows"][:, 0].cpu())
    m1, x1 = O.epe(outs[True]["flows"][:, 0].cpu(), ref["flows"][:, 0])
    m0, x0 = O.epe(outs[False]["flows"][:, 0].cpu(), ref["flows"][:, 0])
    print(f"{kind}: hoisted vs single-chain EPE mean {d_mean:.2e} max {d_max:.2e}; vs the CPU oracle: hoisted {m1:.2e} / {x1:.2e}, "
          f"single-chain {m0:.2e} / {x0:.2e}")
    assert d_mean <= 1e-4 and m1 <= 1e-3 and m0 <= 1e-3
