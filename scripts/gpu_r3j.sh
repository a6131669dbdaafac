#!/bin/bash
# round 3, tenth GPU pass: do forked branches pay once every block is 48 KB (three per CU)?
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/graph_bench.py --batch 1 2>&1 | grep use_graph | tee $O/r3j_graph.log
timeout 300 python scripts/graph_bench.py --batch 1 --swizzled 2>&1 | grep use_graph | tee -a $O/r3j_graph.log
