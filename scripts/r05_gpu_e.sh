#!/bin/bash
# round 5, pass e: fused mask+upsample parity after the rounding fix; in-situ kernel traces + HBM counters of the lookup on both volume layouts
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -x -k "mask_upsample or fused" 2>&1 | tail -8 > $O/r5e_pytest.log; cat $O/r5e_pytest.log | cut -c1-300
timeout 200 python scripts/maskup_bench.py 2>&1 | grep -v Warning | tee $O/r5e_maskup.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1"
tr() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o r -- "$@" > $O/$name.log 2>&1; }
pmc() { name=$1; shift; ctr=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; }
tr r5e_tr_blk $B --steps 3 --warmup 2
PFK_VOLUME_LAYOUT=rowmajor tr r5e_tr_row $B --steps 3 --warmup 2
pmc r5e_pmc_fetch_blk FETCH_SIZE $B --steps 1 --warmup 1
pmc r5e_pmc_write_blk WRITE_SIZE $B --steps 1 --warmup 1
PFK_VOLUME_LAYOUT=rowmajor pmc r5e_pmc_fetch_row FETCH_SIZE $B --steps 1 --warmup 1
cd $R
for t in blk row; do python scripts/callsite_stats.py $O/r5e_tr_$t --title "kernel trace, volume layout $t" | tee -a $O/r5e_callsites.md; done
python scripts/callsite_stats.py $O/r5e_pmc_fetch_blk --pmc $O/r5e_pmc_fetch_blk $O/r5e_pmc_write_blk --title "PMC passes, blocked" | tee -a $O/r5e_callsites.md
python scripts/callsite_stats.py $O/r5e_pmc_fetch_row --pmc $O/r5e_pmc_fetch_row --title "PMC pass, row-major" | tee -a $O/r5e_callsites.md
du -sh $O/r5e_* | tail -8
