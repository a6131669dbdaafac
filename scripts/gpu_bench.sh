#!/bin/bash
# gpurun --timeout 900 -- 'bash scripts/gpu_bench.sh'
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.log
timeout 200 python bench.py --steps 10 --warmup 3 --cpu-forwards 1 --cpu-budget-s 15 > $O/bench_full.log 2>&1
timeout 200 python bench.py --steps 10 --warmup 3 --skip-dead-upsample --no-cpu-baseline --no-roofline > $O/bench_skip.log 2>&1
timeout 200 python bench.py --steps 5 --warmup 2 --batch 4 --no-cpu-baseline > $O/bench_b4.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in pytest_gpu bench_full bench_skip bench_b4; do echo "== $f"; tail -n 2 $O/$f.log | cut -c1-1500; done
