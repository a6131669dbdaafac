#!/bin/bash
# K13 in situ: {pair, fused} x {mask branch on the side stream, on the main stream}
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
B="python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split-modes --no-extra-legs --no-batch1 --no-roofline"
for fuse in 0 1; do for ov in 1 0; do
  PFK_FUSE_MASK=$fuse PFK_OVERLAP=$ov timeout 300 $B 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse=$fuse overlap=$ov:', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a $O/r5i_k13.log
done; done
