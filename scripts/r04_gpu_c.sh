#!/bin/bash
# round 4, pass C: fixed tests with full output, K7 (bf16, launch-wide gate) tests + timings
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_reference_models.py tests/test_gpu_dropin.py tests/test_gpu_capi_stub.py "tests/test_gpu_train_step.py::test_train_step_raft" "tests/test_gpu_train_step.py::test_update_block_backward_ops_exact_on_their_inputs" -m gpu -q -s 2>&1 | grep -v "^\s*$" > gpurun_out/r4c_tests.log
tail -15 gpurun_out/r4c_tests.log
timeout 600 python scripts/lookup_bench.py > gpurun_out/r4c_lookup.log 2>&1
cat gpurun_out/r4c_lookup.log
