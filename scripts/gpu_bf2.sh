#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=${CFG8} --reps 8 --only ${ONLY:-c2,zr1,q1,fm} > $O/convbf_abl.log 2>&1; cat $O/convbf_abl.log
