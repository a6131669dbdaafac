#!/bin/bash
# training path: gradient tests + the train leg of the bench
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py tests/test_gpu_ddp.py tests/test_gpu_corr_bwd.py -m gpu -q --tb=short -x 2>&1 | tail -5 | cut -c1-220
timeout 300 python scripts/train_bench.py 2>&1 | tail -4 | cut -c1-220
timeout 400 python bench.py --steps 5 --no-cpu-baseline --no-split-modes --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'train',d['train']['value'],d['train']['ms_per_step'],d['train']['encoders_fwd_bwd_ms'])"
