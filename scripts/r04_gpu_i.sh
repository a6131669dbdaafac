#!/bin/bash
# round 4, pass I: stream-K schedules at batch 8 (tile quantisation: 1760 / 2640 tiles on 512 / 768 resident blocks)
set -x
mkdir -p gpurun_out
timeout 900 python scripts/conv_bench.py --batch 8 --cfgs=-1,9,35,36,37,38,39,40,41,30,31,32,33,34 --only q1h,cv,c2,zr1h,c1,f2,fm,mk --rounds 3 --reps 10 > gpurun_out/r4i_conv_b8.log 2>&1; grep -v amdgpu gpurun_out/r4i_conv_b8.log | sed 's/err [0-9.e+-]*//g; s/ us / /g' | cut -c1-520
