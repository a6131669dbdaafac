#!/bin/bash
# Round 3: the GPU passes behind profiles/r03_a / r03_b (each was one `gpurun -- bash scripts/r03_gpu_passes.sh <pass>` call;
# logs under gpurun_out/r3<pass>_*).  Usage: bash scripts/r03_gpu_passes.sh b|c|d|e|f|g|h|i|j|k|l|m|n|o|p|q|r|s
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
case "$1" in
b)
  # round 3, second GPU pass: the persistent pipelined kernel — correctness, then schedule sweeps against the round-2 kernels
  timeout 900 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $O/r3b_pytest.log; cat $O/r3b_pytest.log | cut -c1-250
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,53,52,51,61,69 --reps 20 > $O/r3b_conv_b8.log 2>&1; cat $O/r3b_conv_b8.log | cut -c1-400
  timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,4,53,52,51,61,57 --reps 30 > $O/r3b_conv_b1.log 2>&1; cat $O/r3b_conv_b1.log | cut -c1-400
  timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=4,21,22,23,25,26 --only fm,zr1,mk --reps 20 > $O/r3b_conv_abl_b8.log 2>&1; cat $O/r3b_conv_abl_b8.log | cut -c1-400
  timeout 600 python scripts/corr_bench.py 2>&1 | grep "K1" > $O/r3b_corr.log; cat $O/r3b_corr.log
  timeout 600 python -m pytest tests/test_gpu_train_step.py -m gpu -q -s -k "train_step_raft" 2>&1 | grep -E "achieved|passed|failed|worst L2" > $O/r3b_train_gate.log; cat $O/r3b_train_gate.log
  ;;
c)
  # round 3, third GPU pass: where does the fp32 K loop lose its 20 %?  ablation ladder at batch 8 + counters; forked / graph schedules
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
  ;;
d)
  # round 3, fourth GPU pass: tile-shape / pipeline-depth experiments at batch 8, skeletons, pp ablations; eager vs forked vs graph at batch 1
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,4,1,2,3,5,6,7,11,12,13 --only fm,c2,zr1,q1,mk --reps 10 > $O/r3d_conv_tiles_b8.log 2>&1; cat $O/r3d_conv_tiles_b8.log | cut -c1-700
  timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=23,27,28,29,53,82,83,84,52,85,86,87 --only fm --reps 10 > $O/r3d_conv_abl_b8.log 2>&1; cat $O/r3d_conv_abl_b8.log | cut -c1-900
  timeout 300 python scripts/graph_bench.py --batch 1 2>&1 | grep use_graph | tee $O/r3d_graph.log
  ;;
e)
  # round 3, fifth GPU pass: two staging register sets (prefetch distance 2) in the tile and the persistent kernels
  timeout 900 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -8 > $O/r3e_pytest.log; cat $O/r3e_pytest.log | cut -c1-250
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,14,4,15,11,16,12,17,53,88,52,89,61,91 --only fm,c2,zr1,q1,mk,c1 --reps 20 > $O/r3e_conv_b8.log 2>&1; cat $O/r3e_conv_b8.log | cut -c1-900
  timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,4,15,8,52,89,61,91 --reps 30 > $O/r3e_conv_b1.log 2>&1; cat $O/r3e_conv_b1.log | cut -c1-600
  ;;
f)
  # round 3, sixth GPU pass: 64x128 heuristic (interleaved A/B), headline, calibrated K3 traffic
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
  ;;
g)
  # round 3, seventh GPU pass: component ladder of the fp32 K loop (scripts/mfma_probe.hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/mfma_probe.hip -o /tmp/mfma_probe 2>&1 | grep -E "error" 
  timeout 300 /tmp/mfma_probe > $O/r3g_probe.log 2>&1; cat $O/r3g_probe.log
  ;;
h)
  # round 3, eighth GPU pass: counters of the tile kernel vs the persistent kernel on the dominant launch (fm, batch 8)
  cd /tmp && export TMPDIR=/tmp
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3h_pmc_$i -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --cfgs=10,53,11,84 --only fm --reps 3 > $O/r3h_pmc_$i.log 2>&1
    tail -3 $O/r3h_pmc_$i.log | cut -c1-200
  done
  python $GRAFT_REPO_ROOT/scripts/pmc_by_kernel.py $O/r3h_pmc_* --match=conv_gemm > $O/r3h_counters.txt; cat $O/r3h_counters.txt
  ;;
i)
  # round 3, ninth GPU pass: persistent kernel with round-robin whole tiles (L2-sharing) + stream-K remainder
  timeout 900 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -8 > $O/r3i_pytest.log; cat $O/r3i_pytest.log | cut -c1-250
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,53,52,61,69 --reps 10 --rounds 3 > $O/r3i_conv_b8.log 2>&1; cat $O/r3i_conv_b8.log | cut -c1-500
  timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,53,52,61 --reps 20 --rounds 3 > $O/r3i_conv_b1.log 2>&1; cat $O/r3i_conv_b1.log | cut -c1-400
  timeout 600 python scripts/corr_bench.py 2>&1 | grep "K1 fp32" > $O/r3i_corr.log; cat $O/r3i_corr.log
  ;;
j)
  # round 3, tenth GPU pass: do forked branches pay once every block is 48 KB (three per CU)?
  timeout 300 python scripts/graph_bench.py --batch 1 2>&1 | grep use_graph | tee $O/r3j_graph.log
  timeout 300 python scripts/graph_bench.py --batch 1 --swizzled 2>&1 | grep use_graph | tee -a $O/r3j_graph.log
  ;;
k)
  # round 3: L2 behaviour of the persistent kernel after the round-robin tile order
  cd /tmp && export TMPDIR=/tmp
  i=0
  for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3k_pmc_$i -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --cfgs=10,53,69 --only fm --reps 3 > $O/r3k_pmc_$i.log 2>&1
  done
  python $GRAFT_REPO_ROOT/scripts/pmc_by_kernel.py $O/r3k_pmc_* --match=conv_gemm > $O/r3k_counters.txt; cat $O/r3k_counters.txt
  ;;
l)
  # round 3: training step with accumulated weight gradients; B5 seam; full GPU suite; full bench line
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -12 > $O/r3l_pytest.log; cat $O/r3l_pytest.log | cut -c1-250
  timeout 900 python bench.py > $O/r3l_bench.log 2>&1; tail -n 1 $O/r3l_bench.log | python -c "
  import json,sys
  d=json.loads(sys.stdin.read())
  print('value',d['value'],'roofline',d['roofline']['frac'],d['roofline'].get('traffic'))
  for k in ('batch1','model_benchmark_protocol','dropin','config4','train','skip_dead_upsample','split_bf16','cpu_baseline','epe_vs_cpu'): print(k, json.dumps(d.get(k))[:600])
  print('config3', {k:{kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','epe_mean','error','iters4','iters12','err_vs_cpu_fp32')} for k,v in d['config3'].items()})
  "
  ;;
m)
  # round 3: multiplier arithmetic instead of 64-bit divisions at the head of every tile
  timeout 900 python -m pytest tests/test_gpu_conv_pp.py tests/test_gpu_kernels.py tests/test_gpu_conv_fuzz.py tests/test_gpu_splitbf16.py tests/test_gpu_encoder.py -m gpu -q --tb=short -x 2>&1 | tail -5 > $O/r3m_pytest.log; cat $O/r3m_pytest.log | cut -c1-250
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=-1,10,4 --reps 10 --rounds 3 > $O/r3m_conv_b8.log 2>&1; cat $O/r3m_conv_b8.log | cut -c1-300
  timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,4 --reps 20 --rounds 3 > $O/r3m_conv_b1.log 2>&1; cat $O/r3m_conv_b1.log | cut -c1-250
  timeout 600 python bench.py --no-extra-legs --no-cpu-baseline > $O/r3m_bench.log 2>&1; tail -n 1 $O/r3m_bench.log | python -c "
  import json,sys
  d=json.loads(sys.stdin.read()); print('value',d['value'],'roofline',d['roofline']['frac'],d['roofline']['avg_us'],'batch1',d['batch1']['value'],'split',{k:round(v['value'],1) for k,v in d['split_bf16'].items()},'skip',d['skip_dead_upsample']['value'])"
  ;;
n)
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,10010,11,10011 --reps 10 --rounds 5 --only fm,zr1,q1,c2,mk > $O/r3n_conv_b8.log 2>&1; cat $O/r3n_conv_b8.log | cut -c1-330
  timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=4,10004 --reps 20 --rounds 5 --only q1,cv,mk,c1,f2 > $O/r3n_conv_b1.log 2>&1; cat $O/r3n_conv_b1.log | cut -c1-250
  ;;
o)
  # round 3: persistent kernel with the vector-memory counter drained once per tile (counted vmcnt(3) inside the K loop)
  timeout 600 python -m pytest tests/test_gpu_conv_pp.py -m gpu -q --tb=short -x 2>&1 | tail -3
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,53,69,52,11 --reps 10 --rounds 3 > $O/r3o_conv_b8.log 2>&1; cat $O/r3o_conv_b8.log | cut -c1-420
  timeout 600 python scripts/conv_bench.py --batch 1 --cfgs=-1,52,53 --reps 20 --rounds 3 > $O/r3o_conv_b1.log 2>&1; cat $O/r3o_conv_b1.log | cut -c1-300
  ;;
p)
  # round 3: does the shader clock hold under the K loop?  probe with s_memtime / s_memrealtime, and rocm-smi polled under the real fm conv
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/mfma_probe.hip -o /tmp/mfma_probe 2>&1 | grep -E "error"
  timeout 400 /tmp/mfma_probe > $O/r3p_probe.log 2>&1; tail -n 16 $O/r3p_probe.log | cut -c1-200
  ( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|mclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.25; done ) > $O/r3p_smi.log 2>&1 &
  SMI=$!
  timeout 300 python scripts/conv_bench.py --batch 8 --cfgs=10,11 --reps 200 --rounds 3 --only fm > $O/r3p_conv_b8.log 2>&1; cat $O/r3p_conv_b8.log | cut -c1-300
  wait $SMI; sort $O/r3p_smi.log | uniq -c | sort -rn | head -8 | cut -c1-250
  timeout 600 python -m pytest tests/test_gpu_conv_pp.py tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "volume or corr" 2>&1 | tail -3
  ;;
q)
  # round 3: where does the split-bf16 kernel (K8) stand?  tile ladder at batch 8, then counters of fm for bf16x6 128x64, bf16x3 128x128 and fp32
  timeout 600 python scripts/conv_bench.py --batch 8 --cfgs=10,300,301,302,303,200,201,202,203 --reps 10 --rounds 3 --only fm,zr1,c2,q1,mk > $O/r3q_conv_b8.log 2>&1; cat $O/r3q_conv_b8.log | cut -c1-500
  cd /tmp && export TMPDIR=/tmp
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3q_pmc_$i -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --cfgs=10,302,203 --only fm --reps 3 > $O/r3q_pmc_$i.log 2>&1
    tail -3 $O/r3q_pmc_$i.log | cut -c1-200
  done
  python $GRAFT_REPO_ROOT/scripts/pmc_by_kernel.py $O/r3q_pmc_* --match=conv_gemm > $O/r3q_counters.txt; cat $O/r3q_counters.txt
  ;;
r)
  # round 3: counters of fm on the split-bf16 kernels AFTER the eight-wave tiles (heuristic: 128x256 + tail split; two planes: 128x128 x2)
  cd /tmp && export TMPDIR=/tmp
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/r3r_pmc_$i -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py --batch 8 --cfgs=300,200,100 --only fm,zr1 --reps 3 > $O/r3r_pmc_$i.log 2>&1
    tail -3 $O/r3r_pmc_$i.log | cut -c1-200
  done
  python $GRAFT_REPO_ROOT/scripts/pmc_by_kernel.py $O/r3r_pmc_* --match=conv_gemm > $O/r3r_counters.txt; cat $O/r3r_counters.txt
  ;;
s)
  # round 3: shader clock and socket power under each kernel family (rocm-smi polled every 50 ms while one fm convolution loops ~2 s)
  for cfg in 10 300 200 100; do
    ( for i in $(seq 1 70); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.05; done ) > $O/r3s_smi_$cfg.log 2>&1 &
    SMI=$!
    timeout 120 python scripts/conv_bench.py --batch 8 --cfgs=$cfg --reps 3000 --only fm > $O/r3s_conv_$cfg.log 2>&1
    wait $SMI
    echo "cfg $cfg: $(grep -v amdgpu $O/r3s_conv_$cfg.log | head -1 | cut -c1-110)"
    grep -o "sclk clock level: [0-9S]*: ([0-9]*Mhz)" $O/r3s_smi_$cfg.log | sort | uniq -c | sort -rn | head -4
    grep -o "Socket Graphics Package Power (W): [0-9.]*" $O/r3s_smi_$cfg.log | sort -t: -k2 -n | tail -2
  done
  ;;
*) echo "unknown pass $1"; exit 2;;
esac
