#!/bin/bash
# round 4, pass Q: GRU epilogues with their global operands preloaded for the whole wave tile (tile-grid kernel) — timing + parity
set -x
mkdir -p gpurun_out
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10 --only zr1h,zr1hL,q1h,q1hL,zr1,q1 --rounds 5 --reps 10 > gpurun_out/r4q_conv_b8.log 2>&1; grep -v amdgpu gpurun_out/r4q_conv_b8.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_pp.py tests/test_gpu_conv_fuzz.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value', round(d['value'],2), 'ms', round(d['ms_per_step'],2))"
