#!/bin/bash
# round 4, pass J: the persistent kernel's hybrid schedule (whole tiles round-robin + stream-K remainder) on the long-K mid-size grids
set -x
mkdir -p gpurun_out
timeout 900 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,50,51,52,53,54,55,56,57,58,59,61,63 --only c2,cv --rounds 3 --reps 10 > gpurun_out/r4j_conv_b8.log 2>&1; grep -v amdgpu gpurun_out/r4j_conv_b8.log | sed 's/err [0-9.e+-]*//g; s/ us / /g; s/cfg  *//g; s/ TF//g' | cut -c1-560
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,1,5,11,12,0,4 --only mk,c1 --rounds 3 --reps 10 > gpurun_out/r4j_conv_mk.log 2>&1; grep -v amdgpu gpurun_out/r4j_conv_mk.log | sed 's/err [0-9.e+-]*//g; s/ us / /g; s/cfg  *//g; s/ TF//g' | cut -c1-560
