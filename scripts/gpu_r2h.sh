#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python scripts/enc_grad_check.py 2>&1 | grep -v amdgpu.ids | tee $O/r2h_enc.log | cut -c1-250
