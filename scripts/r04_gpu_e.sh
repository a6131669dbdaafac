#!/bin/bash
# round 4, pass E2: batch-1 tile/schedule alternatives for the under-filled launches
set -x
mkdir -p gpurun_out
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,0,4,8,9,10,35,36,37 --only f2,c1,q1,cv,mk,c2,fm,zr1 --rounds 3 > gpurun_out/r4e_conv_b1.log 2>&1
cat gpurun_out/r4e_conv_b1.log | cut -c1-600
