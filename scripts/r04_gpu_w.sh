#!/bin/bash
# pass W: ccmr / ms_raft_plus, stock = the reference's own torch fallback for a missing alt_cuda_corr extension
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; : > $O/r4w_speedup.jsonl
run() { timeout 80 python scripts/dropin_speedup.py "$@" 2>/dev/null | tail -n 1 | tee -a $O/r4w_speedup.jsonl | cut -c1-420; }
run ms_raft_plus ms_raft_plus MSRAFTPlus --stock-without-plugin --n 3 --warm 1
run ccmr ccmr CCMR --stock-without-plugin --n 3 --warm 1
