#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_live_model.py tests/test_gpu_bf16_gate.py::test_bf16_pyramid_and_lookup tests/test_gpu_dropin.py -m gpu -q -s --tb=short 2>&1 | grep -v "^$" > $O/r2d_new.log
grep -n "passed\|failed" $O/r2d_new.log | tail -3
grep -n "Error\|assert \|worst\|   [0-9]\|FAILED\|encoder_train" $O/r2d_new.log | cut -c1-330 | head -70
timeout 300 python - <<'PY' 2>&1 | tail -12
import sys, time, torch
sys.path.insert(0, '.')
from bench import train_leg, timed
dev = torch.device('cuda:0')
import ptlflow_amd; ptlflow_amd.load_native()
r = train_leg(dev)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != 'config'})
PY
