#!/usr/bin/env python3
"""Time the correlation path's kernels on the north-star shape (55x128 grid, D=256, L=4, r=4; GPU box): K1 fp32 (tile walk:
row-major vs 16x16 supertiles), K1 bf16, K2, K3 on fp32 / bf16 pyramids, lookup backward, volume backward."""
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


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n   # us


h, w, D, L, r = 55, 128, 256, 4, 4
N = h * w
for B in (1, 8):
    f1 = torch.randn(B, N, D, device=dev); f2 = torch.randn(B, N, D, device=dev)
    vol = torch.empty(B, N, N, device=dev)
    flop = 2.0 * B * N * N * D
    us = timeit(lambda: ops.corr_volume(f1, f2, 1 / 16.0, vol))
    print(f"K1 fp32 B={B}: {us:.1f} us  {flop/us/1e6:.1f} TFLOP/s ({100*flop/us/1e6/157.3:.1f}% of 157.3)  write {4*B*N*N/us/1e6:.2f} TB/s")
    ref = vol.clone()
    for cfg in (4, 10, 11, 12, 13, 53, 51):   # v3 64x64 padded / swizzled tile grids; 50+v: persistent pipelined kernel (1 swizzled, 2 XCD groups, 4*bpc)
        ops.debug_set_tile(cfg)
        try:
            us = timeit(lambda: ops.corr_volume(f1, f2, 1 / 16.0, vol), n=10)
            print(f"   K1 fp32 B={B} tile cfg {cfg}: {us:.1f} us  {flop/us/1e6:.1f} TFLOP/s  same={bool(torch.equal(vol, ref))}")
        except Exception as e:
            print(f"   cfg {cfg}: {e}")
    ops.debug_set_tile(-1)
    f2b = f2.to(torch.bfloat16)
    volb = torch.empty(B, N, N, device=dev, dtype=torch.bfloat16)
    f1b = f1.to(torch.bfloat16)
    us = timeit(lambda: ops.corr_volume_bf16(f1b, f2b, 1 / 16.0, volb))
    ref = torch.bmm(f1b.float(), f2b.float().transpose(1, 2)) / 16 if B == 1 else ref
    err = (volb.float() - ref).abs().max().item()
    print(f"K1 bf16 B={B}: {us:.1f} us  write {2*B*N*N/us/1e6:.2f} TB/s ({100*2*B*N*N/us/1e6/8:.1f}% of 8 TB/s)  {flop/us/1e6:.0f} TFLOP/s  max|bf16-fp32| {err:.3e}")
    for dt, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        lv, hh, ww = [], h, w
        for l in range(L):
            lv.append(torch.empty(B * N, hh, ww, device=dev, dtype=dt)); hh //= 2; ww //= 2
        lv[0].copy_((ref if dt == torch.float32 else volb).view(B * N, h, w))
        def pools():
            for l in range(1, L):
                ops.corr_pool2x2(lv[l - 1], lv[l])
        us = timeit(pools)
        el = 4 if dt == torch.float32 else 2
        byt = sum(v.numel() for v in lv[:-1]) * el + sum(v.numel() for v in lv[1:]) * el
        print(f"K2 {name} B={B}: {us:.1f} us  {byt/us/1e6:.2f} TB/s")
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        coords = (torch.stack([xs, ys], 0)[None] + torch.randn(B, 2, h, w, device=dev) * 6).contiguous()
        out = torch.empty(B * N, 324, device=dev)
        alg = B * (N * L * (100 * el + 81 * 4) + 8 * N)
        for pix in (4, 8):
            ops.debug_set_lookup_pix(pix)
            us = timeit(lambda: ops.corr_lookup(lv, coords, r, out), n=50)
            ck = float(out.double().sum())
            print(f"K3 {name} B={B} pix={pix}: {us:.1f} us  algorithmic {alg/1e6:.1f} MB -> {alg/us/1e6:.2f} TB/s ({100*alg/us/1e6/8:.1f}% of 8 TB/s)  checksum {ck:.6e}")
        ops.debug_set_lookup_pix(4)
    # the blocked 4 x 8 layout of the inference path (round 5): K1 against the permuted / padded target map (7168 rows for 55 x 128),
    # K2 blocked -> blocked, K3 on the blocked levels (same values: checksum of the lookups must equal the row-major one's)
    nb = ops.blocked_map_elems(h, w)
    f2blk = torch.empty(B * nb, D, device=dev)
    us_p = timeit(lambda: ops.fmap_to_blocked(f2.view(B * N, D), f2blk, B, h, w))
    volk = torch.empty(B, N, nb, device=dev)
    us = timeit(lambda: ops.corr_volume(f1, f2blk.view(B, nb, D), 1 / 16.0, volk))
    print(f"K1 fp32 B={B} blocked target map ({nb} columns): {us:.1f} us  {flop/us/1e6:.1f} TFLOP/s of real work ({100*flop/us/1e6/157.3:.1f}% of 157.3); row permutation {us_p:.1f} us")
    lb, hh, ww, dims = [volk.view(B * N, nb)], h, w, [(h, w)]
    for l in range(1, L):
        hh //= 2; ww //= 2
        lb.append(torch.empty(B * N, ops.blocked_map_elems(hh, ww), device=dev)); dims.append((hh, ww))
    def pools_b():
        for l in range(1, L):
            ops.corr_pool2x2_blocked(lb[l - 1], lb[l], dims[l - 1][0], dims[l - 1][1])
    us = timeit(pools_b)
    byt = sum(v.numel() for v in lb[:-1]) * 4 + sum(v.numel() for v in lb[1:]) * 4
    print(f"K2 fp32 B={B} blocked: {us:.1f} us  {byt/us/1e6:.2f} TB/s")
    out = torch.empty(B * N, 324, device=dev)
    alg = B * (N * L * (100 * 4 + 81 * 4) + 8 * N)
    us = timeit(lambda: ops.corr_lookup_blocked(lb, [d[0] for d in dims], [d[1] for d in dims], coords, r, out), n=50)
    print(f"K3 fp32 B={B} blocked: {us:.1f} us  algorithmic {alg/1e6:.1f} MB -> {alg/us/1e6:.2f} TB/s ({100*alg/us/1e6/8:.1f}% of 8 TB/s)")
    # backward pieces (training shapes are smaller; here the same shape for comparability)
    if B == 1:
        sizes = [(int(v.shape[1]), int(v.shape[2])) for v in lv]
        bufs = [torch.zeros(B * N, (a * b + 3) // 4 * 4, device=dev) for a, b in sizes]
        g = torch.randn(B * N, 324, device=dev)
        us = timeit(lambda: ops.corr_lookup_bwd(bufs, [s[0] for s in sizes], [s[1] for s in sizes], coords, r, g), n=50)
        print(f"K3 backward B={B}: {us:.1f} us")
