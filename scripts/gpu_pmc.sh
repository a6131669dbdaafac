#!/bin/bash
# PMC passes on the conv micro-benchmark (separate runs per counter set; no trace domains besides kernel-trace).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $O/pmc/counters_list.txt 2>&1
CMD="python $GRAFT_REPO_ROOT/scripts/conv_bench.py --only q1,zr1,fm --cfgs=0,4 --reps 5"
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- $CMD > $O/pmc/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE TCP_TCC_READ_REQ_sum
ls -R $O/pmc | head -40
