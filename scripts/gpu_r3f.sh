#!/bin/bash
# round 3, sixth GPU pass: 64x128 heuristic (interleaved A/B), headline, calibrated K3 traffic
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_pp.py tests/test_gpu_kernels.py tests/test_gpu_conv_fuzz.py -m gpu -q --tb=short -x 2>&1 | tail -6 > $O/r3f_pytest.log; cat $O/r3f_pytest.log | cut -c1-250
timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,11,4,12 --reps 10 --rounds 5 > $O/r3f_conv_b8.log 2>&1; cat $O/r3f_conv_b8.log | cut -c1-400
timeout 600 python bench.py --no-extra-legs --no-split-modes --no-cpu-baseline > $O/r3f_bench.log 2>&1; tail -n 1 $O/r3f_bench.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3f_pmc_$tag -- python $GRAFT_REPO_ROOT/scripts/traffic_probe.py > $O/r3f_pmc_$tag.log 2>&1
  tail -1 $O/r3f_pmc_$tag.log | cut -c1-300
done
python $GRAFT_REPO_ROOT/scripts/pmc_by_kernel.py $O/r3f_pmc_* --match=lookup_kernel,pool2x2,direct_copy,copyBuffer > $O/r3f_traffic.txt; cat $O/r3f_traffic.txt
