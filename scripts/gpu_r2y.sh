#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -6 | cut -c1-220
timeout 400 python bench.py --steps 10 --no-cpu-baseline --no-extra-legs > $O/y_bench.log 2>&1; tail -1 $O/y_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'],'roofline',d['roofline']['frac'],'split',{k:round(v['value'],1) for k,v in d.get('split_bf16',{}).items()})
print({k:v['avg_us'] for k,v in d['kernels'].items()})"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/y_pmc_fetch -o p -- $B --steps 1 --warmup 1 > $O/y_pmc_fetch.log 2>&1
python $R/scripts/pmc_extract.py --fetch $O/y_pmc_fetch --write $O/y_pmc_fetch --batch 8 | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read()).items(): print(k,'fetch x2 KB',2*v['fetch_kb'],v['avg_us'])"
