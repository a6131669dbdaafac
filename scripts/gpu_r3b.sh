#!/bin/bash
# round 3, second GPU pass: the persistent pipelined kernel — correctness, then schedule sweeps against the round-2 kernels
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $O/r3b_pytest.log; cat $O/r3b_pytest.log | cut -c1-250
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,53,52,51,61,69 --reps 20 > $O/r3b_conv_b8.log 2>&1; cat $O/r3b_conv_b8.log | cut -c1-400
timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,4,53,52,51,61,57 --reps 30 > $O/r3b_conv_b1.log 2>&1; cat $O/r3b_conv_b1.log | cut -c1-400
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=4,21,22,23,25,26 --only fm,zr1,mk --reps 20 > $O/r3b_conv_abl_b8.log 2>&1; cat $O/r3b_conv_abl_b8.log | cut -c1-400
timeout 600 python scripts/corr_bench.py 2>&1 | grep "K1" > $O/r3b_corr.log; cat $O/r3b_corr.log
timeout 600 python -m pytest tests/test_gpu_train_step.py -m gpu -q -s -k "train_step_raft" 2>&1 | grep -E "achieved|passed|failed|worst L2" > $O/r3b_train_gate.log; cat $O/r3b_train_gate.log
