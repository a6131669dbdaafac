#!/bin/bash
# round 4, pass F: the loop-invariant context hoist — parity, then the bench
set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_seam_model.py tests/test_gpu_live_model.py tests/test_gpu_splitbf16.py -m gpu -q -x -s 2>&1 | grep -v "^\s*$" | tail -40 > gpurun_out/r4f_tests.log
tail -25 gpurun_out/r4f_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r4f_bench.log 2> gpurun_out/r4f_bench.err
tail -c 1500 gpurun_out/r4f_bench.log; tail -3 gpurun_out/r4f_bench.err
