#!/bin/bash
# round 4, pass P: fp32 convolution with the weight operand staged by LDS-DMA (cfg 14 = 64x64 x3, 15 = 64x128 x2) vs register staging (10 / 11)
set -x
mkdir -p gpurun_out
timeout 900 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,14,11,15 --rounds 3 --reps 10 > gpurun_out/r4p_conv_b8.log 2>&1; grep -v amdgpu gpurun_out/r4p_conv_b8.log | sed 's/ us / /g; s/cfg  *//g; s/ TF//g' | cut -c1-300
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,10,14 --only mk,fm,c1 --rounds 3 --reps 30 > gpurun_out/r4p_conv_b1.log 2>&1; grep -v amdgpu gpurun_out/r4p_conv_b1.log | sed 's/ us / /g; s/cfg  *//g; s/ TF//g' | cut -c1-300
