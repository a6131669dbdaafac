#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.log 2>&1
timeout 300 python bench.py --model gma --batch 4 --steps 10 --no-cpu-baseline --no-roofline > $O/bench_gma.log 2>&1
for f in pytest_gpu bench_default bench_gma; do echo "== $f"; tail -n 1 $O/$f.log | cut -c1-3500; done
