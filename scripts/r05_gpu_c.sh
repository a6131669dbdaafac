#!/bin/bash
# round 5, pass c: blocked volume layout (K1/K2/K3) parity + timing; the 128x96 tile in the encoders; whole-forward effect
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropin.py tests/test_gpu_bf16_gate.py tests/test_gpu_encoder.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $O/r5c_pytest.log; cat $O/r5c_pytest.log | cut -c1-300
timeout 300 python scripts/lookup_blocked_bench.py 2>&1 | grep -v Warning | tee $O/r5c_lookup.log
timeout 300 python scripts/conv_bench.py --shapes enc --batch 16 --only l1,l3s,l3 --cfgs=-1,3,1,2,11,13 --reps 5 --rounds 3 2>&1 | grep -v Warning | tee $O/r5c_enc.log
timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split-modes --no-extra-legs > $O/r5c_bench.log 2>&1; tail -n 1 $O/r5c_bench.log | cut -c1-300
