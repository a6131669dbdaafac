#!/bin/bash
# round 5, pass b: tile shapes for the encoders' widths (cout 64 / 96) at fnet's batch-8 size (16 images) and cnet's (8)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for b in 16 8; do
timeout 300 python scripts/conv_bench.py --shapes enc --batch $b --only l1 --cfgs=-1,10,12,16,17 --reps 5 --rounds 3 2>&1 | grep -v Warning | tee -a $O/r5b_enc.log
timeout 300 python scripts/conv_bench.py --shapes enc --batch $b --only l2s,l2,l2d --cfgs=-1,14,15 --reps 5 --rounds 3 2>&1 | grep -v Warning | tee -a $O/r5b_enc.log
timeout 300 python scripts/conv_bench.py --shapes enc --batch $b --only l3s,l3,out --cfgs=-1 --reps 5 --rounds 3 2>&1 | grep -v Warning | tee -a $O/r5b_enc.log
done
