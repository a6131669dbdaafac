#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_fuzz.py tests/test_gpu_splitbf16.py tests/test_gpu_bf16_gate.py tests/test_gpu_encoder.py tests/test_gpu_corr_bwd.py -m gpu -q --tb=short -x 2>&1 | tail -5 | cut -c1-220
timeout 400 python bench.py --steps 10 --no-cpu-baseline --no-extra-legs > $O/v_bench.log 2>&1; tail -1 $O/v_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'],'roofline',d['roofline'],'split',{k:round(v['value'],1) for k,v in d.get('split_bf16',{}).items()})
print({k:v['avg_us'] for k,v in d['kernels'].items()})"
timeout 100 python scripts/corr_bench.py 2>&1 | tail -8 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/v_pmc_write -o p -- $B --steps 1 --warmup 1 > $O/v_pmc_write.log 2>&1
python $R/scripts/pmc_extract.py --fetch $O/v_pmc_write --write $O/v_pmc_write --batch 8 | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read()).items(): print(k,v['write_kb'],v['avg_us'])"
