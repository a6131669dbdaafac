#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py -m gpu -q -s --tb=line 2>&1 | grep -v "^$" > $O/r2f_train.log
grep -n "passed\|failed" $O/r2f_train.log | tail -3
grep -n "worst\|   [0-9]\|encoder_train\|Error" $O/r2f_train.log | cut -c1-250 | head -60
timeout 300 python scripts/corr_bench.py 2>&1 | grep "K3" 
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -3
