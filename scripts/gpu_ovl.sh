#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py tests/test_gpu_live_model.py -m gpu -q --tb=short -x 2>&1 | tail -5 | cut -c1-220
for ov in 1 0; do
PFK_OVERLAP=$ov timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-extra-legs --no-split-modes --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap $ov value',d['value'],'batch1',d['batch1']['value'])"
done
