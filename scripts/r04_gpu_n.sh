#!/bin/bash
# round 4, pass N: K8 with the weight planes by LDS-DMA (pfk_debug_set_tile(180 + t)) against register staging, same tiles
set -x
mkdir -p gpurun_out
timeout 900 python scripts/conv_bench.py --batch 8 --cfgs=300,304,8304,305,8305,306,8306 --only fm,zr1h,q1h,c2,cv,mk --rounds 3 --reps 10 > gpurun_out/r4n_conv_x6.log 2>&1; grep -v amdgpu gpurun_out/r4n_conv_x6.log | sed 's/ us / /g; s/cfg  *//g; s/ TF//g' | cut -c1-400
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=200,206,8206,204,8204,100,106,8106,104,8104 --only fm,zr1h,q1h --rounds 3 --reps 10 > gpurun_out/r4n_conv_x31.log 2>&1; grep -v amdgpu gpurun_out/r4n_conv_x31.log | sed 's/ us / /g; s/cfg  *//g; s/ TF//g' | cut -c1-500
