#!/bin/bash
# K13, co-resident form: parity, standalone timing, in situ 2x2
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -x -k "mask_upsample or fused" 2>&1 | tail -6 | cut -c1-300
timeout 200 python scripts/maskup_bench.py 2>&1 | grep -v Warning | tee $O/r5j_maskup.log
B="python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split-modes --no-extra-legs --no-batch1 --no-roofline"
for fuse in 0 1; do for ov in 1 0; do
  PFK_FUSE_MASK=$fuse PFK_OVERLAP=$ov timeout 300 $B 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse=$fuse overlap=$ov:', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a $O/r5j_k13.log
done; done
