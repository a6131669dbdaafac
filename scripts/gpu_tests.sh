#!/bin/bash
# Run on the GPU box via: gpurun --timeout 900 -- 'bash scripts/gpu_tests.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
