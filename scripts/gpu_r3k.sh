#!/bin/bash
# round 3: L2 behaviour of the persistent kernel after the round-robin tile order
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3k_pmc_$i -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --cfgs=10,53,69 --only fm --reps 3 > $O/r3k_pmc_$i.log 2>&1
done
python $GRAFT_REPO_ROOT/scripts/pmc_by_kernel.py $O/r3k_pmc_* --match=conv_gemm > $O/r3k_counters.txt; cat $O/r3k_counters.txt
