#!/bin/bash
# round 4, pass R: counters of the bf16x6 fm launch with the weight planes by LDS-DMA (default) vs register staging (pfk_debug_set_tile(176))
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r4r_pmc_$tag -- python $R/scripts/conv_bench.py --batch 8 --cfgs=306,7306 --only fm,zr1h --reps 3 > $O/r4r_pmc_$tag.log 2>&1
  tail -2 $O/r4r_pmc_$tag.log | cut -c1-200
done
python $R/scripts/pmc_by_kernel.py $O/r4r_pmc_* --match=conv_gemm_bf_kernel > $O/r4r_counters.txt; cat $O/r4r_counters.txt | cut -c1-160
