#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python scripts/train_bisect.py > $O/r2e_bisect.log 2>&1; tail -60 $O/r2e_bisect.log
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_live_model.py tests/test_gpu_bf16_gate.py::test_bf16_pyramid_and_lookup -m gpu -q --tb=short 2>&1 | tail -25 | cut -c1-300
