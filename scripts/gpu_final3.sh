#!/bin/bash
# round-2 evidence, final tree: whole GPU suite, default bench (all legs), micro-benches, kernel traces (fp32 b8, b1, gma, raft_small,
# bf16x3, bf16, training step) and PMC passes (FETCH_SIZE, WRITE_SIZE, SQ busy) — each PMC pass with --kernel-trace only
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
[ -n "$SKIP_PYTEST" ] || { timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -6 > $O/g_pytest.log; cat $O/g_pytest.log | cut -c1-200; }
timeout 600 python bench.py > $O/g_bench.log 2>&1; tail -n 1 $O/g_bench.log | cut -c1-300
timeout 200 python scripts/corr_bench.py > $O/g_corr.log 2>&1
timeout 200 python scripts/lookup_bench.py > $O/g_lookup.log 2>&1
timeout 200 python scripts/conv_bench.py --batch 1 --cfgs=-1,4 --reps 40 > $O/g_conv_b1.log 2>&1
timeout 200 python scripts/conv_bench.py --batch 8 --cfgs=-1 --reps 20 > $O/g_conv_b8.log 2>&1
timeout 200 python scripts/wgrad_bench.py --variants=0,4 > $O/g_wgrad.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
tr() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$name -o r -- "$@" > $O/$name.log 2>&1; }
tr g_tr_f32 $B --steps 3 --warmup 2
tr g_tr_b1 $B --batch 1 --steps 10 --warmup 3
tr g_tr_gma $B --model gma --batch 4 --steps 3 --warmup 2
tr g_tr_small $B --model raft_small --steps 3 --warmup 2
tr g_tr_x3 $B --conv-precision bf16x3 --steps 3 --warmup 2
tr g_tr_bf16 $B --conv-precision bf16 --steps 3 --warmup 2
tr g_tr_train python $R/scripts/train_prof.py
pmc() { name=$1; shift; ctr=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; }
pmc g_pmc_fetch FETCH_SIZE $B --steps 1 --warmup 1
pmc g_pmc_write WRITE_SIZE $B --steps 1 --warmup 1
pmc g_pmc_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $B --steps 1 --warmup 1
ls $O | grep "^g_" | wc -l
