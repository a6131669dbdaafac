#!/bin/bash
# gpurun --timeout 1200 -- 'bash scripts/gpu_bench.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench.log
python bench.py --steps 10 --warmup 3 --skip-dead-upsample --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_skip.log
python bench.py --steps 5 --warmup 2 --batch 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_b4.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -30
