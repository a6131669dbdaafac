#!/bin/bash
# round 5, pass d: fused mask conv2 + softmax + upsampling (parity, timing, whole forward), blocked layout at batch 1 in situ
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "mask_upsample or blocked or layouts" 2>&1 | tail -15 > $O/r5d_pytest.log; cat $O/r5d_pytest.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -x -k "fused or side_stream or headline" 2>&1 | tail -15 > $O/r5d_pytest2.log; cat $O/r5d_pytest2.log | cut -c1-300
timeout 300 python scripts/maskup_bench.py 2>&1 | grep -v Warning | tee $O/r5d_maskup.log
B="python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split-modes --no-extra-legs"
timeout 600 $B > $O/r5d_bench.log 2>&1; tail -n 1 $O/r5d_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused(auto):', d['value'], d['ms_per_step'], 'batch1', d['batch1']['value'], d.get('batch16',{}).get('value'), {k:v['avg_us'] for k,v in d['kernels'].items()})"
PFK_VOLUME_LAYOUT=rowmajor timeout 600 $B > $O/r5d_bench_rowmajor.log 2>&1; tail -n 1 $O/r5d_bench_rowmajor.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rowmajor volume:', d['value'], d['ms_per_step'], 'batch1', d['batch1']['value'])"
