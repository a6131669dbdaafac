#!/bin/bash
# round 4, pass M: what do the GRU epilogues cost?  (the hoisted launches with their gate epilogue vs the linear one)
set -x
mkdir -p gpurun_out
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10 --only zr1h,zr1hL,q1h,q1hL --rounds 5 --reps 10 > gpurun_out/r4m_conv_b8.log 2>&1; grep -v amdgpu gpurun_out/r4m_conv_b8.log | cut -c1-300
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1 --only zr1h,zr1hL,q1h,q1hL --rounds 5 --reps 30 > gpurun_out/r4m_conv_b1.log 2>&1; grep -v amdgpu gpurun_out/r4m_conv_b1.log | cut -c1-300
