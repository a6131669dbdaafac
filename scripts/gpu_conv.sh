#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --cfgs=4,9 --only q1,q2,cv,f2,c1 > $O/conv_sk2.log 2>&1
cat $O/conv_sk2.log
