#!/bin/bash
# round 5, pass a: the new parity pins + the seam-level dead-work skip on the GPU, then the driver's bench command (all legs)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_live_model.py tests/test_gpu_two_ranks_one_device.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $O/r5a_pytest.log; cat $O/r5a_pytest.log | cut -c1-300
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5a_bench.log 2>&1; tail -n 1 $O/r5a_bench.log | cut -c1-600
