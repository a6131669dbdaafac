#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/corr_bench.py 2>&1 | grep "K3\|K1 bf16"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_corr_bwd.py tests/test_gpu_encoder.py tests/test_gpu_train_step.py tests/test_gpu_live_model.py -m gpu -q --tb=line 2>&1 | tail -6 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-split-modes --no-extra-legs --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',d['value'],'batch1',d['batch1']['value'], 'roofline', d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r02_prof_f32 -o r -- $B --steps 3 --warmup 2 > $O/r02_prof_f32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r02_prof_b1 -o r -- $B --batch 1 --steps 10 --warmup 3 > $O/r02_prof_b1.log 2>&1
ls $O/r02_prof_f32 $O/r02_prof_b1 | head
