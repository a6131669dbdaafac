#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_a.log 2>&1
timeout 300 python bench.py > $O/bench_default.log 2>&1
for f in pytest_gpu bench_a bench_default; do echo "== $f"; tail -n 1 $O/$f.log | cut -c1-900; done
