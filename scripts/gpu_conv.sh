#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python scripts/conv_bench.py --cfgs=4,9 > $O/conv_b1.log 2>&1; cat $O/conv_b1.log
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=0,4 --reps 8 > $O/conv_b8.log 2>&1; cat $O/conv_b8.log
