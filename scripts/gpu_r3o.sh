#!/bin/bash
# round 3: persistent kernel with the vector-memory counter drained once per tile (counted vmcnt(3) inside the K loop)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -3
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,53,69,52,11 --reps 10 --rounds 3 > $O/r3o_conv_b8.log 2>&1; cat $O/r3o_conv_b8.log | cut -c1-420
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,52,53 --reps 20 --rounds 3 > $O/r3o_conv_b1.log 2>&1; cat $O/r3o_conv_b1.log | cut -c1-300
