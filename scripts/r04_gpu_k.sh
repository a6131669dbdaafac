#!/bin/bash
# round 4, pass K: the persistent cross-tile-pipelined kernel on the encoders' short-K convolutions (18 K-steps per tile)
set -x
mkdir -p gpurun_out
timeout 600 python scripts/stage_time.py --batch 8 --enc-tiles 50,51,52,53,66,67,69,10,-1 > gpurun_out/r4k_stage_b8.log 2>&1; grep "encoders with\|forward" gpurun_out/r4k_stage_b8.log
timeout 600 python scripts/stage_time.py --batch 1 --enc-tiles 51,53,69,10,-1 > gpurun_out/r4k_stage_b1.log 2>&1; grep "encoders with\|forward" gpurun_out/r4k_stage_b1.log
