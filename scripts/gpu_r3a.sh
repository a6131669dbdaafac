#!/bin/bash
# round 3, first GPU pass: the whole GPU suite (new seam / two-rank / stub / 12-iteration tests), the full bench line with the
# new legs, and batch-8 schedule / ablation sweeps of the fp32 conv kernel that the persistent-kernel work starts from
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -15 > $O/r3a_pytest.log; cat $O/r3a_pytest.log | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_train_step.py -m gpu -q -s -k "train_step_raft" 2>&1 | grep -v "^$" | tail -30 > $O/r3a_train_gate.log
timeout 900 python bench.py --torch-baseline > $O/r3a_bench.log 2>&1; tail -n 1 $O/r3a_bench.log | cut -c1-3000
timeout 300 python scripts/conv_bench.py --batch 8 --cfgs -1,4,10,21,22,23,25,26 --only fm,zr1,mk,q1 --reps 20 > $O/r3a_conv_b8.log 2>&1; cat $O/r3a_conv_b8.log
