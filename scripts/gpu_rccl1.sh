#!/bin/bash
# the torchrun / RCCL branch of bench.py with one rank (all a single-GPU box can do), then the plain invocation for comparison
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-split-modes > $O/rccl1.log 2>&1; tail -2 $O/rccl1.log | cut -c1-400
