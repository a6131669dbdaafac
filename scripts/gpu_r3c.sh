#!/bin/bash
# round 3, third GPU pass: where does the fp32 K loop lose its 20 %?  ablation ladder at batch 8 + counters; forked / graph schedules
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=4,21,22,23,24,25,26,10 --only fm,c2,mk --reps 20 > $O/r3c_conv_abl_b8.log 2>&1; cat $O/r3c_conv_abl_b8.log | cut -c1-500
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -x 2>&1 | tail -8 > $O/r3c_pytest.log; cat $O/r3c_pytest.log | cut -c1-250
for b in 1; do
timeout 300 python scripts/graph_bench.py --batch $b 2>&1 | grep use_graph | tee -a $O/r3c_graph.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD|GRBM|SPI)_[A-Z0-9_]+" | sort -u > $O/r3c_counters.txt; wc -l $O/r3c_counters.txt
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3c_pmc_$tag -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --cfgs=10,4 --only fm --reps 3 > $O/r3c_pmc_$tag.log 2>&1
  tail -2 $O/r3c_pmc_$tag.log | cut -c1-300
done
ls $O | grep r3c
