#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --batch 8 --only mk,c2,fm,c1,f2,cv --cfgs=-1,15,18,19 --reps 10 --rounds 3 2>&1 | grep -v Warning | tee $O/r5h_conv.log
timeout 300 python scripts/conv_bench.py --shapes enc --batch 16 --only l1,l3,out --cfgs=-1,18,19 --reps 5 --rounds 3 2>&1 | grep -v Warning | tee -a $O/r5h_conv.log
