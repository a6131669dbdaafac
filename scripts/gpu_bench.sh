#!/bin/bash
# gpurun --timeout 900 -- 'bash scripts/gpu_bench.sh'
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_gpu.log 2>&1
PFK_CUDNN_BENCHMARK=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_cudnnbench.log 2>&1
PFK_CHANNELS_LAST=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_cl.log 2>&1
for f in pytest_gpu bench_gpu bench_cudnnbench bench_cl; do echo "== $f"; tail -n 2 $O/$f.log | cut -c1-400; done
