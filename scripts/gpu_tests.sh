#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 100 python scripts/lookup_bench.py 2>&1 | tail -5 | tee $O/lookup_bench.log
