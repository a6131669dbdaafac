#!/bin/bash
# round 3, seventh GPU pass: component ladder of the fp32 K loop (scripts/mfma_probe.hip)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/mfma_probe.hip -o /tmp/mfma_probe 2>&1 | grep -E "error" 
timeout 300 /tmp/mfma_probe > $O/r3g_probe.log 2>&1; cat $O/r3g_probe.log
