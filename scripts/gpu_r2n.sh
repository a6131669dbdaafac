#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/corr_bench.py 2>&1 | grep "K3"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py tests/test_gpu_train_step.py -m gpu -q --tb=line 2>&1 | tail -4 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-split-modes --no-extra-legs --no-roofline --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'])"
