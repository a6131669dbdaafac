#!/bin/bash
# round 3, ninth GPU pass: persistent kernel with round-robin whole tiles (L2-sharing) + stream-K remainder
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -8 > $O/r3i_pytest.log; cat $O/r3i_pytest.log | cut -c1-250
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,53,52,61,69 --reps 10 --rounds 3 > $O/r3i_conv_b8.log 2>&1; cat $O/r3i_conv_b8.log | cut -c1-500
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,53,52,61 --reps 20 --rounds 3 > $O/r3i_conv_b1.log 2>&1; cat $O/r3i_conv_b1.log | cut -c1-400
timeout 600 python scripts/corr_bench.py 2>&1 | grep "K1 fp32" > $O/r3i_corr.log; cat $O/r3i_corr.log
