#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,10010,11,10011 --reps 10 --rounds 5 --only fm,zr1,q1,c2,mk > $O/r3n_conv_b8.log 2>&1; cat $O/r3n_conv_b8.log | cut -c1-330
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=4,10004 --reps 20 --rounds 5 --only q1,cv,mk,c1,f2 > $O/r3n_conv_b1.log 2>&1; cat $O/r3n_conv_b1.log | cut -c1-250
