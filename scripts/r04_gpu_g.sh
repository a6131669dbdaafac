#!/bin/bash
# round 4, pass G: the training-path hoist (gradients vs float64), train leg; tile choices for the hoisted GRU launches
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train.py tests/test_gpu_ddp.py tests/test_gpu_two_ranks_one_device.py -m gpu -q -x -s 2>&1 | grep -v "^\s*$" | grep "achieved\|worst L2\|passed\|failed\|Error\|error\|checks" > gpurun_out/r4g_tests.log
cat gpurun_out/r4g_tests.log
timeout 600 python scripts/train_bench.py > gpurun_out/r4g_train.log 2>&1; tail -5 gpurun_out/r4g_train.log
python - > gpurun_out/r4g_trainleg.log 2>&1 <<'PY'
import json, torch, bench
print(json.dumps(bench.train_leg(torch.device("cuda", 0))))
PY
tail -2 gpurun_out/r4g_trainleg.log | cut -c1-600
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,11,12,4 --only zr1h,q1h,zr2h,q2h,zr1,q1 --rounds 3 > gpurun_out/r4g_conv_b8.log 2>&1; cut -c1-420 gpurun_out/r4g_conv_b8.log
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,4,9,10,0 --only zr1h,q1h,zr2h,q2h --rounds 3 > gpurun_out/r4g_conv_b1.log 2>&1; cut -c1-420 gpurun_out/r4g_conv_b1.log
