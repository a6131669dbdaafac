#!/usr/bin/env python3
"""HBM-side read traffic of the pyramid lookup (K3), calibrated: run under
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ...   (one counter group per pass)
next to two kernels whose traffic is known exactly — the 2x2 pooling of the level-0 volume (reads every line of it once,
coalesced) and a plain device copy — so that bytes = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B can be checked
before it is trusted for the lookup's 4-byte gathers (scripts/pmc_extract.py --probe sums the counters per kernel name)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")
torch.manual_seed(0)
B, h, w, L, r = 8, 55, 128, 4, 4
N = h * w
lv, hh, ww = [], h, w
for l in range(L):
    lv.append(torch.randn(B * N, hh, ww, device=dev)); hh //= 2; ww //= 2
ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
out = torch.empty(B * N, 324, device=dev)
big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)      # evicts L2 + MALL between the probes
src = torch.randn(64 << 20, device=dev)                          # 256 MiB copy: calibration kernel 2
dst = torch.empty_like(src)
for rep in range(3):
    coords = (torch.stack([xs, ys], 0)[None] + torch.randn(B, 2, h, w, device=dev) * 6).contiguous()
    big.zero_()
    ops.corr_pool2x2(lv[0], lv[1])
    big.zero_()
    dst.copy_(src)
    big.zero_()
    ops.corr_lookup(lv, coords, r, out)
torch.cuda.synchronize()
alg = B * (N * L * (100 + 81) * 4 + 8 * N)
print(f"known: pool2x2 reads {B*N*54*128*4} B (54 of 55 rows of every level-0 map), copy reads {src.numel()*4} B, "
      f"lookup algorithmic {alg} B ({B*N*L*100*4} B of unique patch reads + {B*N*324*4} B written)")
