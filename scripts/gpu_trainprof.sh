#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_train -o t -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --steps 2 > $O/prof_train.log 2>&1
grep libpfk $O/prof_train.log
