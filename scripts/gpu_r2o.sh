#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_fuzz.py tests/test_gpu_splitbf16.py tests/test_gpu_train.py tests/test_gpu_model.py tests/test_gpu_dropin.py -m gpu -q --tb=line 2>&1 | tail -8 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'],'roofline',d['roofline']['frac'], {k:v['avg_us'] for k,v in d['kernels'].items()}, {k:(round(v['value'],1),v['epe_mean']) for k,v in d['split_bf16'].items()})"
