#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/encprof
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encprof -o enc -- python $GRAFT_REPO_ROOT/scripts/stage_time.py --batch 8 --reps 3 > $O/encprof/run.log 2>&1
tail -2 $O/encprof/run.log
ls $O/encprof
