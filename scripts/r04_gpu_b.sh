#!/bin/bash
# round 4, pass B: remaining new tests; conv phase-stagger / priority experiment
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_reference_models.py tests/test_gpu_two_ranks_one_device.py tests/test_gpu_train_step.py -m gpu -q -s 2>&1 | grep -v "^\s*$" | tail -150 > gpurun_out/r4b_tests.log
tail -30 gpurun_out/r4b_tests.log
export PFK_DEBUG_KNOBS=1
python - > gpurun_out/r4b_stagger.log 2>&1 <<'PY'
import os, sys, math, torch
sys.path.insert(0, os.getcwd())
import ptlflow_amd
from ptlflow_amd.packing import pack_conv_weight
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")
torch.manual_seed(0)
B, H, W = 8, 55, 128
M = B * H * W
ws = torch.zeros(ops.conv_workspace_bytes(), device=dev, dtype=torch.uint8)
def bench(name, segs, cout, kh, kw, settings, reps=20, rounds=3):
    cin = sum(segs)
    xs = [torch.randn(M, c, device=dev) for c in segs]
    wt = torch.randn(cout, cin, kh, kw, device=dev) / math.sqrt(cin * kh * kw)
    bias = torch.randn(cout, device=dev) * 0.1
    offs, o = [], 0
    for c in segs:
        offs.append((o, c, c)); o += c
    packed = pack_conv_weight(wt, offs)
    out = torch.zeros(M, cout, device=dev)
    flops = 2.0 * M * cout * kh * kw * cin
    def run():
        ops.conv2d(xs, B, H, W, kh, kw, packed, bias, cout, 0, False, 1.0, out, None, None, None, ws)
    res = {s: [] for s in settings}
    base = None
    for r in range(rounds):
        for s in settings:
            cfg, stag, prio = s
            ops.debug_set_tile(-1); ops.debug_set_tile(400 + stag); ops.debug_set_tile(500 + prio)
            if cfg >= 0: ops.debug_set_tile(cfg)
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            res[s].append(1e3 * e0.elapsed_time(e1) / reps)
            if base is None: base = out.clone()
            assert torch.equal(base, out), s
    for s in settings:
        v = sorted(res[s]); us = v[len(v)//2]
        print(f"{name} cfg {s[0]:3d} stagger {s[1]:2d} prio {s[2]}: {us:7.1f} us  {flops/us/1e6:6.1f} TF  ({flops/us/1e6/157.3:.3f})", flush=True)
    ops.debug_set_tile(-1); ops.debug_set_tile(400); ops.debug_set_tile(500)
settings = [(-1,0,0),(-1,2,0),(-1,4,0),(-1,6,0),(-1,9,0),(-1,12,0),(-1,0,1),(-1,9,1),(10,0,0),(10,2,0),(10,4,0),(10,5,0),(10,0,1),(4,0,0),(4,4,0),(4,8,0)]
bench("fm", [128], 512, 3, 3, settings)
bench("zr1", [384], 256, 1, 5, [(-1,0,0),(-1,4,0),(-1,8,0),(-1,12,0),(-1,0,1),(10,0,0),(10,4,0)])
bench("c2", [256], 192, 3, 3, [(-1,0,0),(-1,2,0),(-1,4,0),(-1,6,0),(-1,0,1)])
bench("mk", [256], 576, 1, 1, [(-1,0,0),(-1,1,0),(-1,2,0),(-1,0,1)])
PY
cat gpurun_out/r4b_stagger.log
