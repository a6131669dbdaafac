#!/bin/bash
# round 4, pass O: K8 with LDS-DMA weight planes as the default: parity of every tile configuration, bf16 legs
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_splitbf16.py tests/test_gpu_bf16_gate.py tests/test_gpu_conv_fuzz.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r4o_tests.log; cat gpurun_out/r4o_tests.log
python - > gpurun_out/r4o_legs.log 2>&1 <<'PY'
import os, time, torch, json
os.environ["PFK_DEBUG_KNOBS"] = "1"
import ptlflow_amd
from ptlflow_amd.raft import RAFT, GMA
from ptlflow_amd.synth import smooth_pair
ptlflow_amd.load_native()
dev = torch.device("cuda", 0)
x = {"images": smooth_pair(8, 436, 1024, seed=1234).to(dev)}
def t(m, knob):
    torch.ops.pfk.debug_set_tile(knob)
    for _ in range(2): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m(x)
    torch.cuda.synchronize(); return 8 * 5 / (time.perf_counter() - t0)
for prec in ("bf16x6", "bf16x3", "bf16"):
    m = RAFT(iters=32, conv_precision=prec).load_synthetic(1234).eval().to(dev)
    res = {}
    for rnd in range(2):
        for name, knob in (("dma (default)", 100), ("register staging", 170)):
            res.setdefault(name, []).append(round(t(m, knob), 2))
    print(prec, res, flush=True)
    del m; torch.cuda.empty_cache()
torch.ops.pfk.debug_set_tile(100)
PY
cat gpurun_out/r4o_legs.log | grep -v amdgpu
