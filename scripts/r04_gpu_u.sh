#!/bin/bash
# pass U: stock PyTorch-ROCm vs accelerate() on the reference's own classes (one process per model, each under its own timeout)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; : > $O/r4u_speedup.jsonl
run() { timeout 60 python scripts/dropin_speedup.py "$@" 2>/dev/null | tail -n 1 | tee -a $O/r4u_speedup.jsonl | cut -c1-330; }
run raft raft RAFT
run gma gma GMA
run lcv lcv_raft LCV_RAFT
run raft raft RAFTSmall
run sea_raft sea_raft SEARAFT --kw '{"block_dims": [64, 128, 256]}'
run skflow skflow SKFlow
run rapidflow rapidflow RAPIDFlow

run ms_raft_plus ms_raft_plus MSRAFTPlus --stock-kw '{"alternate_corr": false}'

