#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x 2>&1 | tail -3
timeout 200 python scripts/stage_time.py --batch 8
timeout 200 python scripts/stage_time.py --batch 8 --conv-precision bf16x3
timeout 200 python scripts/stage_time.py --batch 8 --conv-precision bf16x6
timeout 200 python scripts/stage_time.py --batch 1
timeout 200 python scripts/stage_time.py --batch 1 --conv-precision bf16x3
