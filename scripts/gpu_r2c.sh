#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_corr_bwd.py tests/test_gpu_live_model.py tests/test_gpu_bf16_gate.py tests/test_gpu_train_step.py tests/test_gpu_encoder.py tests/test_gpu_model.py -m gpu -q -s --tb=short 2>&1 | grep -v "^$" > $O/r2c_new.log
grep -n "passed\|failed" $O/r2c_new.log | tail -3
grep -n "Error\|assert \|worst\|   [0-9]\|gap\|FAILED" $O/r2c_new.log | cut -c1-300 | head -80
timeout 300 python scripts/corr_bench.py > $O/r2c_corr.log 2>&1; tail -20 $O/r2c_corr.log
