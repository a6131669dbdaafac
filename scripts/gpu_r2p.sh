#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 200 python scripts/corr_bench.py 2>&1 | grep "K1"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r02_prof_train -o r -- python $GRAFT_REPO_ROOT/scripts/train_prof.py > $O/r02_prof_train.log 2>&1
tail -2 $O/r02_prof_train.log
python $GRAFT_REPO_ROOT/scripts/trace_stats.py $O/r02_prof_train --forwards 4 --top 40 | cut -c1-210
