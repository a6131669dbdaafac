#!/bin/bash
# round 3, fifth GPU pass: two staging register sets (prefetch distance 2) in the tile and the persistent kernels
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -8 > $O/r3e_pytest.log; cat $O/r3e_pytest.log | cut -c1-250
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,14,4,15,11,16,12,17,53,88,52,89,61,91 --only fm,c2,zr1,q1,mk,c1 --reps 20 > $O/r3e_conv_b8.log 2>&1; cat $O/r3e_conv_b8.log | cut -c1-900
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,4,15,8,52,89,61,91 --reps 30 > $O/r3e_conv_b1.log 2>&1; cat $O/r3e_conv_b1.log | cut -c1-600
