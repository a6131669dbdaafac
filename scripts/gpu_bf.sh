#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_splitbf16.py -m gpu -q -x -s 2>&1 | tail -15
timeout 300 python scripts/conv_bench.py --cfgs=-1,101,102,103 > $O/convbf_b1.log 2>&1; cat $O/convbf_b1.log
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=-1,101,102,103 --reps 8 > $O/convbf_b8.log 2>&1; cat $O/convbf_b8.log
