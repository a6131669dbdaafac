#!/bin/bash
# gpurun --timeout 900 -- 'bash scripts/gpu_bench.sh'
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os,bench;print('cpu_count',os.cpu_count(),'affinity',len(os.sched_getaffinity(0)),'host_cores',bench.host_cores())"; lscpu | grep -E "Model name|^CPU\(s\)"; } > $O/host.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_gpu.log 2>&1
timeout 200 python bench.py --steps 10 --warmup 3 --skip-dead-upsample --no-cpu-baseline --no-roofline > $O/bench_skip.log 2>&1
timeout 200 python bench.py --steps 5 --warmup 2 --batch 4 --no-cpu-baseline > $O/bench_b4.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 150 python bench.py --steps 5 --warmup 2 --cpu-forwards 1 --cpu-budget-s 20 > $O/bench_full.log 2>&1
tail -2 $O/*.log
ls -R $O/prof | head -20
