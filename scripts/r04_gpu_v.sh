#!/bin/bash
# pass V: lookups as plain NCHW for torch consumers — the three foreign-consumer families again, then their tests
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; : > $O/r4v_speedup.jsonl
run() { timeout 60 python scripts/dropin_speedup.py "$@" 2>/dev/null | tail -n 1 | tee -a $O/r4v_speedup.jsonl | cut -c1-330; }
run skflow skflow SKFlow
run sea_raft sea_raft SEARAFT --kw '{"block_dims": [64, 128, 256]}'
run rapidflow rapidflow RAPIDFlow
timeout 200 python -m pytest tests/test_gpu_reference_siblings.py tests/test_gpu_reference_models.py -m gpu -q -x -k "sibling or sea_raft" --tb=short 2>&1 | tail -4 | cut -c1-300
