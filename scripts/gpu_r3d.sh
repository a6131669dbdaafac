#!/bin/bash
# round 3, fourth GPU pass: tile-shape / pipeline-depth experiments at batch 8, skeletons, pp ablations; eager vs forked vs graph at batch 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,4,1,2,3,5,6,7,11,12,13 --only fm,c2,zr1,q1,mk --reps 10 > $O/r3d_conv_tiles_b8.log 2>&1; cat $O/r3d_conv_tiles_b8.log | cut -c1-700
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=23,27,28,29,53,82,83,84,52,85,86,87 --only fm --reps 10 > $O/r3d_conv_abl_b8.log 2>&1; cat $O/r3d_conv_abl_b8.log | cut -c1-900
timeout 300 python scripts/graph_bench.py --batch 1 2>&1 | grep use_graph | tee $O/r3d_graph.log
