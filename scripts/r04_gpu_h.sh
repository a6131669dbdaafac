#!/bin/bash
# round 4, pass H: after the hoist — side-stream overlap on/off, batch 4/8/16; split-bf16 tile choices on the hoisted GRU shapes
set -x
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1 --steps 10 --warmup 3"
for tag in "b8" "b8_nooverlap" "b16" "b4"; do
  case $tag in
    b8) $B > gpurun_out/r4h_$tag.log 2>/dev/null ;;
    b8_nooverlap) PFK_OVERLAP=0 $B > gpurun_out/r4h_$tag.log 2>/dev/null ;;
    b16) $B --batch 16 > gpurun_out/r4h_$tag.log 2>/dev/null ;;
    b4) $B --batch 4 > gpurun_out/r4h_$tag.log 2>/dev/null ;;
  esac
  python -c "
import json,sys
for l in open('gpurun_out/r4h_$tag.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$tag', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')"
done
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=300,303,304,305,306,301,302 --only zr1h,q1h,zr2h,q2h --rounds 3 > gpurun_out/r4h_conv_bf.log 2>&1; grep -v amdgpu gpurun_out/r4h_conv_bf.log | sed 's/err [0-9.e+-]*//g' | cut -c1-400
