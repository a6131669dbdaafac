#!/bin/bash
# pass S: K3 with cross-lane tap reads against the LDS patch (timing + counters), then the final evidence on the rebuilt tree
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "lookup" --tb=short 2>&1 | tail -5
timeout 300 python scripts/lookup_bench.py 2>&1 | grep "^lookup" | tee $O/r4s_lookup.txt
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lk.py <<'PY'
import os, sys, torch
os.environ["PFK_DEBUG_KNOBS"] = "1"
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ptlflow_amd; ptlflow_amd.load_native(); ops = torch.ops.pfk
dev = torch.device("cuda"); torch.manual_seed(0)
B, h, w, L, r = 8, 55, 128, 4, 4; N = h * w
lv, hh, ww = [], h, w
for l in range(L):
    lv.append(torch.randn(B * N, hh, ww, device=dev)); hh //= 2; ww //= 2
ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], 0)[None] + torch.randn(B, 2, h, w, device=dev) * 6).contiguous()
out = torch.empty(B * N, 324, device=dev)
for v in (4, 14, 4, 14):
    ops.debug_set_lookup_pix(v)
    for _ in range(3): ops.corr_lookup(lv, coords, r, out)
torch.cuda.synchronize()
PY
for c in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_BUSY_CYCLES" "FETCH_SIZE" ; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r4s_pmc_$n -o p -- python /tmp/lk.py > $O/r4s_pmc_$n.log 2>&1
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4s_tr -o r -- python /tmp/lk.py > $O/r4s_tr.log 2>&1
cd $R
SKIP_PYTEST=1 bash scripts/gpu_final_r04.sh 2>&1 | tail -4
