#!/bin/bash
# Compile pfk_gemm.hip to ISA and report, per kernel, the things that silently wreck performance:
# scratch use, waterfall loops (v_readfirstlane next to buffer ops), s_load inside loops, instruction mix.
cd /tmp/isa && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -I/root/repo/ptlflow_amd/csrc -save-temps -c /root/repo/ptlflow_amd/csrc/pfk_gemm.hip -o g.o 2>&1 | grep -E "error" 
S=pfk_gemm-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - "$S" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
# split per kernel
parts = re.split(r'\n(_ZN[^\n:]*conv_gemm[^\n:]*):', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i+1]
    body = body.split('s_endpgm')[0]
    m = re.search(r'conv_gemm(_v3)?_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d)E', name)
    tag = f"{'v3' if m.group(1) else 'v1'} {m.group(2)}x{m.group(3)} epi{m.group(4)}"
    c = lambda pat: len(re.findall(pat, body))
    print(f"{tag:18s} scratch={c(r'scratch_')} readfirstlane={c(r'v_readfirstlane')} s_load={c(r's_load_')} "
          f"buffer_load={c(r'buffer_load')} mfma={c(r'v_mfma')} ds_read={c(r'ds_read')} ds_write={c(r'ds_write')} lines={body.count(chr(10))}")
PY
