#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --cfgs=4,5,8 > $O/conv_b1.log 2>&1
timeout 300 python scripts/conv_bench.py --batch 4 --cfgs=1,5,6 --reps 10 > $O/conv_b4.log 2>&1
cat $O/conv_b1.log $O/conv_b4.log
