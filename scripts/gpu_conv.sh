#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=0,1,4,3 --reps 8 > $O/conv_b8.log 2>&1
cat $O/conv_b8.log
