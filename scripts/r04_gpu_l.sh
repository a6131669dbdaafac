#!/bin/bash
set -x
mkdir -p gpurun_out
PFK_DEBUG_KNOBS=1 timeout 300 python scripts/enc_prof.py --batch 1 --tile 10 > gpurun_out/r4l_enc_b1_t10.log 2>&1
PFK_DEBUG_KNOBS=1 timeout 300 python scripts/enc_prof.py --batch 1 > gpurun_out/r4l_enc_b1.log 2>&1
grep "==" gpurun_out/r4l_enc_b1_t10.log gpurun_out/r4l_enc_b1.log
