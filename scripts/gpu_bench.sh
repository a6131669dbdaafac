#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.log
timeout 100 python scripts/host_overhead.py > $O/host_overhead.log 2>&1
timeout 300 python bench.py > $O/bench_default.log 2>&1
for f in pytest_gpu host_overhead bench_default; do echo "== $f"; tail -n 4 $O/$f.log | cut -c1-2500; done
