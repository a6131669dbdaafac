#!/bin/bash
# round 2, call B: seam fixes + corr/upsample backward + train step + bf16 corr path — new tests first, then the whole GPU suite and the bench
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_corr_bwd.py tests/test_gpu_live_model.py tests/test_gpu_bf16_gate.py tests/test_gpu_train_step.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 > $O/r2b_new.log
cat $O/r2b_new.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_corr_bwd.py --deselect tests/test_gpu_live_model.py --deselect tests/test_gpu_bf16_gate.py --deselect tests/test_gpu_train_step.py 2>&1 | tail -15 > $O/r2b_all.log
cat $O/r2b_all.log | cut -c1-300
timeout 600 python bench.py > $O/r2b_bench.log 2>&1
tail -n 2 $O/r2b_bench.log | cut -c1-7000
