#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_dropin.py -m gpu -q --tb=short -x 2>&1 | tail -5 | cut -c1-220
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
trc() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$name -o r -- "$@" > $O/$name.log 2>&1; grep -o '"value": [0-9.]*' $O/$name.log | head -1; }
trc x_tr_f32 $B --steps 3 --warmup 2
trc x_tr_b1 $B --batch 1 --steps 10 --warmup 3
cd $R
python scripts/trace_stats.py $O/x_tr_f32 --forwards 5 --top 8 | grep -i "flow_delta\|lookup\|convex"
python scripts/trace_stats.py $O/x_tr_b1 --forwards 13 --top 12 | grep -i "flow_delta\|lookup\|convex"
