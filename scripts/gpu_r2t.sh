#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -m gpu -q --tb=short -x 2>&1 | tail -5 | cut -c1-220
timeout 300 python scripts/conv_bench.py --batch 1 --cfgs=-1,0,4,10,8,34,35,36,37,39,41,31 --reps 40 > $O/s_conv_b1.log 2>&1; tail -14 $O/s_conv_b1.log | cut -c1-700
timeout 200 python scripts/wgrad_bench.py --variants=0,4,2,1 > $O/s_wgrad.log 2>&1; cat $O/s_wgrad.log | cut -c1-300
