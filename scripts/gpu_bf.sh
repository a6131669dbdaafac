#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_splitbf16.py -m gpu -q -x -s 2>&1 | tail -8
timeout 300 python scripts/conv_bench.py --cfgs=${CFG1:-201,301,302} > $O/convbf_b1.log 2>&1; cat $O/convbf_b1.log
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=${CFG8:-101,103,201,202,203,302,303} --reps 8 > $O/convbf_b8.log 2>&1; cat $O/convbf_b8.log
