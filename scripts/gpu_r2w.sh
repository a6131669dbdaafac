#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -x 2>&1 | tail -8 | cut -c1-220
timeout 400 python bench.py --steps 10 --no-cpu-baseline --no-split-modes > $O/w_bench.log 2>&1; tail -1 $O/w_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'],'roofline',d['roofline']['frac'])
print('protocol',d.get('model_benchmark_protocol',{}).get('value'),'train',d.get('train'))"
