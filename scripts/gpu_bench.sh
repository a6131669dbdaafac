#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_gpu.log 2>&1
timeout 200 python bench.py --steps 5 --warmup 2 --batch 4 --no-cpu-baseline --no-roofline > $O/bench_b4.log 2>&1
timeout 200 python bench.py --steps 4 --warmup 2 --batch 8 --no-cpu-baseline --no-roofline > $O/bench_b8.log 2>&1
timeout 200 python bench.py --steps 3 --warmup 1 --batch 16 --no-cpu-baseline --no-roofline > $O/bench_b16.log 2>&1
for f in pytest_gpu bench_gpu bench_b4 bench_b8 bench_b16; do echo "== $f"; tail -n 1 $O/$f.log | cut -c1-330; done
