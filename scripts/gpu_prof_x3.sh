#!/bin/bash
# rocprofv3 kernel trace of the bench command in bf16x3 mode (csv, per-kernel)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_x3 -o r -- python $GRAFT_REPO_ROOT/bench.py --conv-precision bf16x3 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-batch1 --no-split-modes > $O/prof_x3.log 2>&1
tail -1 $O/prof_x3.log | cut -c1-300
ls $O/prof_x3
