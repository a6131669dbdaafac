#!/bin/bash
# round 2, call A: seam fixes + corr/upsample backward + train step — new tests first, then the whole GPU suite and the bench
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_corr_bwd.py tests/test_gpu_live_model.py tests/test_gpu_train_step.py -m gpu -q -x 2>&1 | tail -40 > $O/r2a_new.log
cat $O/r2a_new.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r2a_all.log
cat $O/r2a_all.log
timeout 600 python bench.py > $O/r2a_bench.log 2>&1
tail -n 2 $O/r2a_bench.log | cut -c1-6000
