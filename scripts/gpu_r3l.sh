#!/bin/bash
# round 3: training step with accumulated weight gradients; B5 seam; full GPU suite; full bench line
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -12 > $O/r3l_pytest.log; cat $O/r3l_pytest.log | cut -c1-250
timeout 900 python bench.py > $O/r3l_bench.log 2>&1; tail -n 1 $O/r3l_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'roofline',d['roofline']['frac'],d['roofline'].get('traffic'))
for k in ('batch1','model_benchmark_protocol','dropin','config4','train','skip_dead_upsample','split_bf16','cpu_baseline','epe_vs_cpu'): print(k, json.dumps(d.get(k))[:600])
print('config3', {k:{kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','epe_mean','error','iters4','iters12','err_vs_cpu_fp32')} for k,v in d['config3'].items()})
"
