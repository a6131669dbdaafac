#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 100 python scripts/conv_bench.py --only c2,zr1 --cfgs=4,9 2>&1 | tail -4
