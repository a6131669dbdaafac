#!/bin/bash
# round 3: multiplier arithmetic instead of 64-bit divisions at the head of every tile
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_pp.py tests/test_gpu_kernels.py tests/test_gpu_conv_fuzz.py tests/test_gpu_splitbf16.py tests/test_gpu_encoder.py -m gpu -q --tb=short -x 2>&1 | tail -5 > $O/r3m_pytest.log; cat $O/r3m_pytest.log | cut -c1-250
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,4 --reps 10 --rounds 3 > $O/r3m_conv_b8.log 2>&1; cat $O/r3m_conv_b8.log | cut -c1-300
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,4 --reps 20 --rounds 3 > $O/r3m_conv_b1.log 2>&1; cat $O/r3m_conv_b1.log | cut -c1-250
timeout 600 python bench.py --no-extra-legs --no-cpu-baseline > $O/r3m_bench.log 2>&1; tail -n 1 $O/r3m_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'roofline',d['roofline']['frac'],d['roofline']['avg_us'],'batch1',d['batch1']['value'],'split',{k:round(v['value'],1) for k,v in d['split_bf16'].items()},'skip',d['skip_dead_upsample']['value'])"
