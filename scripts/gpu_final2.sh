#!/bin/bash
# round-2 evidence: whole GPU suite, default bench (all legs), kernel traces (fp32 b8, b1, bf16x3, bf16, training step) and PMC passes
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -6 > $O/f2_pytest.log; cat $O/f2_pytest.log | cut -c1-200
timeout 600 python bench.py > $O/f2_bench.log 2>&1; tail -n 1 $O/f2_bench.log | cut -c1-400
timeout 200 python scripts/corr_bench.py > $O/f2_corr.log 2>&1
timeout 200 python scripts/lookup_bench.py > $O/f2_lookup.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
tr() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$name -o r -- "$@" > $O/$name.log 2>&1; }
tr f2_tr_f32 $B --steps 3 --warmup 2
tr f2_tr_b1 $B --batch 1 --steps 10 --warmup 3
tr f2_tr_x3 $B --conv-precision bf16x3 --steps 3 --warmup 2
tr f2_tr_bf16 $B --conv-precision bf16 --steps 3 --warmup 2
tr f2_tr_train python $R/scripts/train_prof.py
pmc() { name=$1; shift; ctr=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; }
pmc f2_pmc_fetch FETCH_SIZE $B --steps 1 --warmup 1
pmc f2_pmc_write WRITE_SIZE $B --steps 1 --warmup 1
pmc f2_pmc_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $B --steps 1 --warmup 1
pmc f2_pmc_fetch_bf16 FETCH_SIZE $B --conv-precision bf16 --steps 1 --warmup 1
pmc f2_pmc_write_bf16 WRITE_SIZE $B --conv-precision bf16 --steps 1 --warmup 1
ls $O | grep f2_ | head -40
