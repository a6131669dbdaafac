#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py -m gpu -q -s --tb=line -k "norm_forward or encoder_train" 2>&1 | grep -v "^$" > $O/r2g_norm.log
grep -n "passed\|failed" $O/r2g_norm.log | tail -3
grep -n "d x relative\|encoder_train\|Error" $O/r2g_norm.log | cut -c1-250 | head -40
timeout 300 python scripts/corr_bench.py 2>&1 | grep "K1"
timeout 600 python -m pytest tests/test_gpu_bf16_gate.py -m gpu -q -s --tb=short 2>&1 | tail -12 | cut -c1-250
