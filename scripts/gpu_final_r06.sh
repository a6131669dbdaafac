#!/bin/bash
# round-6 evidence, final tree: whole GPU suite (with the slowest tests listed), smoke, the driver's bench command (all legs),
# micro-benches, kernel traces (fp32 batch 8 / batch 1, the K8b bf16-storage forward at batch 8 / batch 1, the seam path on the
# reference's own class, the training step) and PMC passes (FETCH_SIZE, WRITE_SIZE, the SQ busy set — fp32 and bf16 forwards), each
# PMC pass with --kernel-trace only.  One `gpurun -- bash scripts/gpu_final_r06.sh` call; tables by scripts/make_profiles_r06.py (CPU).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
[ -n "$SKIP_PYTEST" ] || { timeout 1800 python -m pytest tests -m gpu -q --tb=line --durations=15 2>&1 | tail -30 > $O/z6_pytest.log; tail -3 $O/z6_pytest.log | cut -c1-200; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/z6_bench.log 2>&1; tail -n 1 $O/z6_bench.log | cut -c1-300
timeout 200 python scripts/corr_bench.py > $O/z6_corr.log 2>&1
timeout 300 python scripts/lookup_blocked_bench.py > $O/z6_lookup_blocked.log 2>&1
timeout 200 python scripts/maskup_bench.py > $O/z6_maskup.log 2>&1
timeout 200 python scripts/maskup_b16_bench.py > $O/z6_maskup_b16.log 2>&1
timeout 200 python scripts/cin2_bench.py > $O/z6_cin2.log 2>&1
timeout 200 python scripts/batch1_check.py > $O/z6_batch1.log 2>&1
timeout 200 python scripts/batch1_check.py --skip-dead > $O/z6_batch1_skip.log 2>&1
timeout 200 python scripts/conv_bench.py --batch 8 --cfgs=-1 --reps 10 --rounds 3 > $O/z6_conv_b8.log 2>&1
timeout 200 python scripts/enc_time.py > $O/z6_enc_time.log 2>&1
timeout 200 python scripts/stage_time.py --conv-precision bf16 --batch 8 > $O/z6_stage_bf16.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
tr() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o r -- "$@" > $O/$name.log 2>&1; }
tr z6_tr_f32 $B --steps 3 --warmup 2
tr z6_tr_b1 $B --batch 1 --steps 10 --warmup 3
tr z6_tr_bf16 $B --conv-precision bf16 --steps 5 --warmup 3
tr z6_tr_bf16_b1 $B --conv-precision bf16 --batch 1 --steps 10 --warmup 3
tr z6_tr_seam python $R/scripts/seam_prof.py
tr z6_tr_train python $R/scripts/train_prof.py
pmc() { name=$1; shift; ctr=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; }
pmc z6_pmc_fetch FETCH_SIZE $B --steps 1 --warmup 1
pmc z6_pmc_write WRITE_SIZE $B --steps 1 --warmup 1
pmc z6_pmc_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $B --steps 1 --warmup 1
pmc z6_pmc_fetch_bf16 FETCH_SIZE $B --conv-precision bf16 --steps 1 --warmup 1
pmc z6_pmc_write_bf16 WRITE_SIZE $B --conv-precision bf16 --steps 1 --warmup 1
pmc z6_pmc_sq_bf16 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $B --conv-precision bf16 --steps 1 --warmup 1
ls $O | grep "^z6_" | wc -l
