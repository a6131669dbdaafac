#!/bin/bash
# round 4, pass D: K7 gate re-timed; where the encoders' time goes (batch 1 / 8)
set -x
mkdir -p gpurun_out
timeout 600 python scripts/lookup_bench.py 2>&1 | grep "altcorr" > gpurun_out/r4d_lookup.log
cat gpurun_out/r4d_lookup.log | cut -c1-400
for b in 1 8; do
  timeout 600 python scripts/stage_time.py --batch $b > gpurun_out/r4d_stage_b$b.log 2>&1; tail -3 gpurun_out/r4d_stage_b$b.log
  timeout 600 python scripts/enc_prof.py --batch $b > gpurun_out/r4d_enc_b$b.log 2>&1; grep "==" gpurun_out/r4d_enc_b$b.log
done
