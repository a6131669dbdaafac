#!/bin/bash
# round-5 evidence, final tree: whole GPU suite, smoke, the driver's bench command (all legs), micro-benches, kernel traces (fp32 b8, b1,
# the seam path on the reference's own class, training step) and PMC passes (FETCH_SIZE, WRITE_SIZE, SQ busy) — each PMC pass with
# --kernel-trace only.  One `gpurun -- bash scripts/gpu_final_r05.sh` call; tables by scripts/make_profiles_r05.py (CPU).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
[ -n "$SKIP_PYTEST" ] || { timeout 1800 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -8 > $O/z_pytest.log; cat $O/z_pytest.log | cut -c1-200; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/z_bench.log 2>&1; tail -n 1 $O/z_bench.log | cut -c1-400
timeout 200 python scripts/corr_bench.py > $O/z_corr.log 2>&1
timeout 300 python scripts/lookup_blocked_bench.py > $O/z_lookup_blocked.log 2>&1
timeout 200 python scripts/maskup_bench.py > $O/z_maskup.log 2>&1
timeout 200 python scripts/conv_bench.py --batch 1 --cfgs=-1,4 --reps 40 > $O/z_conv_b1.log 2>&1
timeout 200 python scripts/conv_bench.py --batch 8 --cfgs=-1,10 --reps 10 --rounds 3 > $O/z_conv_b8.log 2>&1
timeout 300 python scripts/conv_bench.py --shapes enc --batch 16 --cfgs=-1,10,14,15 --reps 5 --rounds 3 > $O/z_conv_enc.log 2>&1
timeout 200 python scripts/enc_time.py > $O/z_enc_time.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
tr() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o r -- "$@" > $O/$name.log 2>&1; }
tr z_tr_f32 $B --steps 3 --warmup 2
tr z_tr_b1 $B --batch 1 --steps 10 --warmup 3
tr z_tr_seam python $R/scripts/seam_prof.py
tr z_tr_train python $R/scripts/train_prof.py
pmc() { name=$1; shift; ctr=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; }
pmc z_pmc_fetch FETCH_SIZE $B --steps 1 --warmup 1
pmc z_pmc_write WRITE_SIZE $B --steps 1 --warmup 1
pmc z_pmc_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $B --steps 1 --warmup 1
PFK_VOLUME_LAYOUT=rowmajor pmc z_pmc_fetch_row FETCH_SIZE $B --steps 1 --warmup 1
PFK_VOLUME_LAYOUT=rowmajor tr z_tr_f32_row $B --steps 3 --warmup 2
ls $O | grep "^z_" | wc -l
