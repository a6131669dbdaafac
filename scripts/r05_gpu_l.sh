#!/bin/bash
# cout_active (mask half of the fused head GEMM skipped with the full launch's bits): parity + what it buys the skip-dead legs
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_live_model.py tests/test_gpu_conv_fuzz.py -m gpu -q --tb=short -x -k "cout_active or skip_dead or fused or conv_linear or fuzz" 2>&1 | tail -6 | cut -c1-300
timeout 200 python scripts/conv_bench.py --batch 1 --only fm --cfgs=-1 --reps 40 2>&1 | grep "^fm"
timeout 900 python3 - <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
import bench, ptlflow_amd
from ptlflow_amd.raft import RAFT
from ptlflow_amd.synth import smooth_pair
ptlflow_amd.load_native()
dev = torch.device("cuda:0")
m = RAFT(iters=32).load_synthetic(1234).eval()
cpu_state = {k: v.clone() for k, v in m.state_dict().items()}
pair = smooth_pair(1, 436, 1024, seed=1234)
b8 = {"images": smooth_pair(8, 436, 1024, seed=1234).to(dev)}
with torch.no_grad():
    leg = bench.dropin_leg(cpu_state, dev, 436, 1024, 32, pair, None, False, batch8=b8)
print(json.dumps({k: leg[k] for k in ("value", "ms_median", "batch8", "skip_dead")}))
for every in (True, False):
    mm = RAFT(iters=32, upsample_every_iter=every).load_synthetic(1234).eval().to(dev)
    one = {"images": pair.to(dev)}
    with torch.no_grad():
        t1 = bench.timed(lambda: mm(one), 3, 10); t8 = bench.timed(lambda: mm(b8), 2, 5)
    print(f"mirror upsample_every_iter={every}: batch 1 {1/t1:.2f} pairs/s ({1e3*t1:.2f} ms), batch 8 {8/t8:.2f} ({1e3*t8:.2f} ms)")
PY
