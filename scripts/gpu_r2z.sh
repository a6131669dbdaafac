#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,4,36,37,41 --reps 20 > $O/z_conv_b8.log 2>&1; tail -13 $O/z_conv_b8.log | cut -c1-400
