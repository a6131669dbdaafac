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
