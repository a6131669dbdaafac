#!/bin/bash
# round 4, pass A: the new reference-class tests + plumbing, then the bench
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_live_model.py tests/test_gpu_reference_models.py tests/test_gpu_two_ranks_one_device.py tests/test_gpu_train_step.py -m gpu -q -x -s 2>&1 | tail -150 > gpurun_out/r4a_tests.log
tail -40 gpurun_out/r4a_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r4a_bench.log 2> gpurun_out/r4a_bench.err
tail -c 3000 gpurun_out/r4a_bench.log; tail -5 gpurun_out/r4a_bench.err
