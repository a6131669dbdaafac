#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_fuzz.py tests/test_gpu_model.py tests/test_gpu_dropin.py tests/test_gpu_train.py -m gpu -q --tb=short -x 2>&1 | tail -5 | cut -c1-220
timeout 400 python bench.py --steps 10 > $O/u_bench.log 2>&1; tail -1 $O/u_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'],'roofline',d['roofline']['frac'], 'cpu', d['cpu_baseline']['value']);
print('protocol',d.get('model_benchmark_protocol'),'\nconfig3',d.get('config3'),'\ntrain',d.get('train'),'\nsplit',{k:round(v['value'],1) for k,v in d.get('split_bf16',{}).items()})
print({k:v['avg_us'] for k,v in d['kernels'].items()})"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/u_pmc_wrreq -o p -- $B --steps 1 --warmup 1 > $O/u_pmc_wrreq.log 2>&1
tail -3 $O/u_pmc_wrreq.log | cut -c1-300
ls $O/u_pmc_wrreq | head
