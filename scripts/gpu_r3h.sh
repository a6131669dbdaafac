#!/bin/bash
# round 3, eighth GPU pass: counters of the tile kernel vs the persistent kernel on the dominant launch (fm, batch 8)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3h_pmc_$i -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --cfgs=10,53,11,84 --only fm --reps 3 > $O/r3h_pmc_$i.log 2>&1
  tail -3 $O/r3h_pmc_$i.log | cut -c1-200
done
python $GRAFT_REPO_ROOT/scripts/pmc_by_kernel.py $O/r3h_pmc_* --match=conv_gemm > $O/r3h_counters.txt; cat $O/r3h_counters.txt
