#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --only c2,cv,fm --cfgs=4,23,25,26 > $O/conv_abl.log 2>&1
cat $O/conv_abl.log
