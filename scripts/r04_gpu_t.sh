#!/bin/bash
# pass T: sibling families on the GPU, the stream-K work-per-CU rule A/B (batch 1 and the training crop), final evidence
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 380 python -m pytest tests/test_gpu_reference_siblings.py -m gpu -q -s --tb=short 2>&1 | grep -v Warning | tail -12 | cut -c1-400 | tee $O/r4t_siblings.txt
timeout 200 python scripts/conv_bench.py --batch 1 --cfgs=-1,4,9 --reps 40 --rounds 3 --only zr1h,zr2h,q1h,c2,fm,f2 2>&1 | grep -v amdgpu.ids | tee $O/r4t_conv_b1.txt
timeout 200 python scripts/conv_bench.py --batch 10 --H 46 --W 62 --cfgs=-1,4,9,10 --reps 20 --rounds 3 --only f2,cv,q1h,q1,c2,zr1h 2>&1 | grep -v amdgpu.ids | tee $O/r4t_conv_train.txt
cd $R
SKIP_PYTEST=1 bash scripts/gpu_final_r04.sh 2>&1 | tail -4
