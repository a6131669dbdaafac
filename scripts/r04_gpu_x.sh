#!/bin/bash
# pass X: LCV-RAFT with its learnable volume behind seam B1 — parity on the real classes, then stock vs accelerated
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_reference_siblings.py -m gpu -q -s --tb=short -k "lcv" 2>&1 | grep -v Warn | grep "EPE\|passed\|failed\|Error\|assert" | cut -c1-300 | tee $O/r4x_lcv.txt
timeout 40 python scripts/dropin_speedup.py lcv lcv_raft LCV_RAFT 2>/dev/null | tail -n 1 | tee $O/r4x_speedup.jsonl | cut -c1-400
