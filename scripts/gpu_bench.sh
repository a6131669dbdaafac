#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_default.log 2>&1
timeout 300 python bench.py --batch 16 --steps 8 --no-cpu-baseline --no-roofline --no-batch1 > $O/bench_b16.log 2>&1
timeout 300 python bench.py --model gma --batch 4 --steps 5 --cpu-forwards 1 > $O/bench_gma.log 2>&1
for f in pytest_gpu bench_default bench_b16 bench_gma; do echo "== $f"; tail -n 2 $O/$f.log | cut -c1-1800; done
