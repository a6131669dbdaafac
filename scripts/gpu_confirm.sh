#!/bin/bash
# last confirmation of a tree: the whole GPU suite, smoke(), the default bench line
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -6 > $O/c_pytest.log; cat $O/c_pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/c_bench.log 2>&1; tail -n 1 $O/c_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'],'roofline',d['roofline']['frac'],'proto',d['model_benchmark_protocol']['value'],'train',d['train']['value'],'split',{k:round(v['value'],1) for k,v in d['split_bf16'].items()},'c3',{k:round(v['value'],1) for k,v in d['config3'].items()})"
