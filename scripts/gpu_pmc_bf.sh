#!/bin/bash
# PMC passes on the split-bf16 conv micro-benchmark (separate runs per counter set; kernel-trace only).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/pmcbf
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --only ${ONLY:-fm,zr1} --cfgs=${CFGS:-203,302} --reps 3"
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmcbf/$name -o $name -- $CMD > $O/pmcbf/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
run sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA
ls $O/pmcbf/*
