#!/bin/bash
# end-of-round evidence: GPU tests, default bench, batch-1 bench, rocprofv3 kernel traces (fp32 default + bf16x3) and PMC passes
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.log 2>&1
timeout 300 python bench.py --batch 1 --steps 30 --no-cpu-baseline --no-roofline > $O/bench_b1.log 2>&1
for f in pytest_gpu bench_default bench_b1; do echo "== $f"; tail -n 1 $O/$f.log | cut -c1-3000; done
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-batch1 --no-split-modes"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_f32 -o r -- $B --steps 3 --warmup 2 > $O/prof_f32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_x3 -o r -- $B --conv-precision bf16x3 --steps 3 --warmup 2 > $O/prof_x3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --steps 1 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --steps 1 --warmup 1 > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o s -- $B --steps 1 --warmup 1 > $O/pmc_sq.log 2>&1
ls $O/prof_f32 $O/prof_x3 $O/pmc_fetch $O/pmc_write $O/pmc_sq
