#!/bin/bash
# full GPU suite + default bench (all legs)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -25 > $O/r2l_all.log
cat $O/r2l_all.log | cut -c1-300
timeout 600 python bench.py > $O/r2l_bench.log 2>&1
tail -n 1 $O/r2l_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('value','ms_per_step','batch1','roofline','cpu_baseline','epe_vs_cpu','split_bf16','skip_dead_upsample','model_benchmark_protocol','config3','train'):
    print(k, json.dumps(d.get(k))[:600])
"
