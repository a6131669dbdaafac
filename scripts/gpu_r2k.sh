#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/enc_grad_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tee $O/r2k_enc.log | awk 'NR%1==0' | head -60
timeout 900 python -m pytest tests/test_gpu_train_step.py -m gpu -q -s --tb=line 2>&1 | grep -v "^$" > $O/r2k_train.log
grep -n "passed\|failed" $O/r2k_train.log | tail -3
grep -n "worst\|   [0-9]\|encoder_train\|Error" $O/r2k_train.log | cut -c1-260 | head -40
