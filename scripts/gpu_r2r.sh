#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_fuzz.py tests/test_gpu_splitbf16.py tests/test_gpu_train.py tests/test_gpu_encoder.py -m gpu -q --tb=line 2>&1 | tail -3 | cut -c1-200
timeout 200 python bench.py --no-cpu-baseline --no-extra-legs --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'],'roofline',d['roofline']['frac'], {k:v['avg_us'] for k,v in d['kernels'].items()}, {k:(round(v['value'],1)) for k,v in d['split_bf16'].items()})"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/f4_pmc_write -o p -- $B --steps 1 --warmup 1 > $O/f4_pmc_write.log 2>&1
python $R/scripts/pmc_extract.py --fetch $O/f4_pmc_write --write $O/f4_pmc_write --batch 8 | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read()).items(): print(k,v['write_kb'],v['avg_us'])"
