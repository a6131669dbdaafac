#!/bin/bash
# rocprofv3 kernel trace of the default bench command + HBM traffic counters for the conv kernels (separate passes)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b8 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-batch1 > $O/prof_b8.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-batch1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $CMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $CMD > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o s -- $CMD > $O/pmc_sq.log 2>&1
ls $O/prof_b8 $O/pmc_fetch $O/pmc_write $O/pmc_sq
