"""Host-side weight re-layout: a CPU emulation of the kernel's K enumeration (source -> tap -> 32-channel
chunk, zero padded) against F.conv2d proves the packed matrix is what pfk_conv2d_f32 expects."""
import torch
import torch.nn.functional as F

from ptlflow_amd.packing import pack_cin2_weight, pack_conv_weight, pack_flow_head_weight, round_up


def emulate(srcs, chans_buf, packed, kh, kw, H, W):
    """srcs: list of [M, C_buf] pixel-major CPU tensors; mirrors the device K loop."""
    M = srcs[0].shape[0]
    cols = []
    for s, cb in zip(srcs, chans_buf):
        img = s.view(1, H, W, cb).permute(0, 3, 1, 2)
        pad = F.pad(img, (kw // 2, kw // 2, kh // 2, kh // 2))
        for ky in range(kh):
            for kx in range(kw):
                tap = pad[:, :, ky:ky + H, kx:kx + W].permute(0, 2, 3, 1).reshape(M, cb)
                cols.append(F.pad(tap, (0, round_up(cb, 32) - cb)))
    A = torch.cat(cols, 1)
    assert A.shape[1] == packed.shape[1]
    return A @ packed.t()


def test_pack_matches_conv2d_multi_source():
    torch.manual_seed(0)
    H, W = 6, 7
    ca, cb_real, cb_buf, cout = 96, 146, 148, 40
    a = torch.randn(1, ca, H, W)
    b = torch.randn(1, cb_real, H, W)
    w = torch.randn(cout, ca + cb_real, 3, 3)
    ref = F.conv2d(torch.cat([a, b], 1), w, padding=1)
    packed = pack_conv_weight(w, [(0, ca, ca), (ca, cb_real, cb_buf)])
    assert packed.shape == (cout, 9 * 96 + 9 * 160)
    a_pm = a.permute(0, 2, 3, 1).reshape(H * W, ca)
    b_pm = F.pad(b.permute(0, 2, 3, 1).reshape(H * W, cb_real), (0, 2))
    out = emulate([a_pm, b_pm], [ca, cb_buf], packed, 3, 3, H, W)
    assert torch.allclose(out.view(1, H, W, cout).permute(0, 3, 1, 2), ref, atol=1e-3)


def test_pack_separable_taps():
    torch.manual_seed(1)
    H, W, c, cout = 5, 9, 64, 32
    x = torch.randn(1, c, H, W)
    for kh, kw in ((1, 5), (5, 1)):
        w = torch.randn(cout, c, kh, kw)
        ref = F.conv2d(x, w, padding=(kh // 2, kw // 2))
        out = emulate([x.permute(0, 2, 3, 1).reshape(H * W, c)], [c], pack_conv_weight(w, [(0, c, c)]), kh, kw, H, W)
        assert torch.allclose(out.view(1, H, W, cout).permute(0, 3, 1, 2), ref, atol=1e-3)


def test_small_kernel_layouts():
    w = torch.arange(128 * 2 * 49, dtype=torch.float32).view(128, 2, 7, 7)
    p = pack_cin2_weight(w)
    assert p.shape == (49, 2, 128) and p[3 * 7 + 2, 1, 5] == w[5, 1, 3, 2]
    w = torch.arange(2 * 256 * 9, dtype=torch.float32).view(2, 256, 3, 3)
    p = pack_flow_head_weight(w)
    assert p.shape == (9, 2, 256) and p[1 * 3 + 2, 1, 77] == w[1, 77, 1, 2]


def test_split_bf16_planes_reconstruct_the_weight():
    """plane 0 = bf16(w), plane k = bf16(residual): the planes' sum approaches w by ~2^-8 per plane, every plane is the
    round-to-nearest of what is left, and a bf16-representable weight needs exactly one plane."""
    from ptlflow_amd.packing import split_bf16_planes
    torch.manual_seed(1)
    w = torch.randn(24, 96) * 0.3
    prev = None
    for n in (1, 2, 3):
        planes = split_bf16_planes(w, n)
        assert planes.dtype == torch.bfloat16 and tuple(planes.shape) == (n, 24, 96)
        err = (w - planes.float().sum(0)).abs().max().item()
        assert err <= float(w.abs().max()) * 2.0 ** (-8 * n) , (n, err)
        if prev is not None:
            assert torch.equal(planes[: n - 1], prev)          # adding a plane never changes the earlier ones
        prev = planes
    exact = split_bf16_planes(w.bfloat16().float(), 2)
    assert torch.equal(exact[0].float(), w.bfloat16().float()) and bool((exact[1] == 0).all())


def test_unpack_wgrad_is_the_inverse_of_pack():
    """The weight gradient comes back from the kernel in the packed [cout, ktot] layout of the forward weight;
    train._unpack_wgrad must undo pack_conv_weight exactly (multi-source, padded channels, cout padded to 4)."""
    from ptlflow_amd.train import _Geometry, _unpack_wgrad
    torch.manual_seed(2)
    cout, kh, kw = 6, 1, 5
    segs = [(0, 96, 96), (96, 146, 148)]
    w = torch.randn(cout, 96 + 146, kh, kw)
    packed = pack_conv_weight(w, segs)
    padded = F.pad(packed, (0, 0, 0, 2))                       # the kernel's output has cout rounded up to a multiple of 4
    back = _unpack_wgrad(padded, w.shape, segs, _Geometry(1, 4, 4, kh, kw))
    assert torch.equal(back, w)


def test_hoisted_gru_packing_equals_the_single_chain_convolution():
    """The loop-invariant hoist of UpdateEngine (ptlflow_amd/update.py): with hx = [h | inp | motion | pad] the engine packs, per
    GRU pass, (a) the weight over the h and motion channels for the per-iteration launch — sources hx[:, :Ch] and hx[:, Ch+Ci:] —
    and (b) the weight over the context channels + the bias for the once-per-forward launch.  Emulating both launches' K loops
    on the CPU, (a) + (b) must equal F.conv2d over cat([h, inp, motion]) with the full weight and bias (raft/update.py:60-62),
    for the SepConvGRU's 1x5 / 5x1 kernels and raft_small's 3x3 ConvGRU (hx padded to a multiple of 4)."""
    from ptlflow_amd.update import UpdateEngine, basic_spec, small_spec
    from ptlflow_amd.synth import synth_state_dict, update_block_shapes
    torch.manual_seed(3)
    H, W = 5, 6
    M = H * W
    for spec in (basic_spec(), small_spec()):
        P = synth_state_dict(update_block_shapes(spec), 5)
        eng = UpdateEngine(P, spec, torch.device("cpu"), "fp32", hoist_context=True)
        Ch, Ci, real, hxc = spec.hidden, spec.context, spec.hidden + spec.x_channels, spec.hx_channels
        hx = torch.randn(M, hxc)
        hx[:, real:] = 0.0                                                  # the pad channels are zero in the engine's buffer
        x_nchw = hx[:, :real].view(1, H, W, real).permute(0, 3, 1, 2)
        for kh, kw, sfx in spec.gru_passes:
            wz, wr, wq = (P[f"gru.conv{k}{sfx}.weight"] for k in "zrq")
            bz, br, bq = (P[f"gru.conv{k}{sfx}.bias"] for k in "zrq")
            for key, wfull, bfull in (("zr", torch.cat([wz, wr], 0), torch.cat([bz, br])), ("q", wq, bq)):
                ref = F.conv2d(x_nchw, wfull, bfull, padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1).reshape(M, -1)
                rest = hx[:, Ch + Ci:]
                per_iter = emulate([hx[:, :Ch], rest], [Ch, rest.shape[1]], eng.w[f"{key}{sfx}.w"], kh, kw, H, W)
                ctx = emulate([hx[:, Ch:Ch + Ci]], [Ci], eng.w[f"{key}c{sfx}.w"], kh, kw, H, W) + eng.w[f"{key}c{sfx}.b"]
                assert eng.w.get(f"{key}{sfx}.b") is None                  # the bias lives in the context term
                assert torch.allclose(per_iter + ctx, ref, atol=2e-4), (key, sfx, float((per_iter + ctx - ref).abs().max()))


def test_permute_mask_head_row_order():
    """`packing.permute_mask_head` = the row order `pfk_mask_upsample_f32` documents (include/pfk.h): row q*160 + j*32 + c is mask
    channel k*64 + s with tap k = 2j + (c >> 4) and sub-pixel s = q*16 + (c & 15); the tenth tap's rows are zero; every one of the
    576 channels appears exactly once."""
    from ptlflow_amd.packing import mask_upsample_perm, permute_mask_head
    w = torch.arange(576, dtype=torch.float32)[:, None].repeat(1, 32) + 1.0      # row r holds the value r + 1
    b = torch.arange(576, dtype=torch.float32) + 1.0
    wp, bp = permute_mask_head(w, b)
    assert wp.shape == (640, 32) and bp.shape == (640,)
    seen = []
    for q in range(4):
        for j in range(5):
            for c in range(32):
                k, s = 2 * j + (c >> 4), q * 16 + (c & 15)
                row = q * 160 + j * 32 + c
                if k == 9:
                    assert float(bp[row]) == 0.0 and bool((wp[row] == 0).all())
                else:
                    assert float(bp[row]) == k * 64 + s + 1.0 and bool((wp[row] == k * 64 + s + 1.0).all())
                    seen.append(k * 64 + s)
    assert sorted(seen) == list(range(576))
    idx, valid = mask_upsample_perm()
    assert int(valid.sum()) == 576 and idx.shape == (640,)
