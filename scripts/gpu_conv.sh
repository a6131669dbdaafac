#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "stream_k or gru or engine" 2>&1 | tail -15 > $O/pytest_sk.log
cat $O/pytest_sk.log
timeout 300 python scripts/conv_bench.py --cfgs=4,9 > $O/conv_b1.log 2>&1
cat $O/conv_b1.log
