#!/bin/bash
# lab aid: registers, scratch and accumulator-half moves outside the asm statements of the one-wave-per-SIMD flash kernel
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -S --cuda-device-only /root/repo/vrdone_amd/csrc/vrd_attn_x3.hip -o /tmp/attn_x3_new.s 2>&1 | grep -v "warning\|^$" | head -20
python3 - <<'PY'
import re
s=open('/tmp/attn_x3_new.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S*w64\S*)(.*?)\.end_amdhsa_kernel', s, re.S):
    b=m.group(2)
    print(m.group(1)[:60], 'vgpr', re.findall(r'next_free_vgpr (\d+)',b), 'sgpr', re.findall(r'next_free_sgpr (\d+)',b), 'scratch bytes', re.findall(r'private_segment_fixed_size (\d+)',b))
for m in re.finditer(r'^(_ZN\S*w64\S*):\n(.*?)\.Lfunc_end\d+:', s, re.S|re.M):
    body=m.group(2); inasm=False; bad=0
    for line in body.split('\n'):
        if 'ASMSTART' in line: inasm=True
        elif 'ASMEND' in line: inasm=False
        elif not inasm and 'v_accvgpr' in line: bad+=1
    print(m.group(1)[:60],'accvgpr outside asm',bad,'scratch instrs',len(re.findall(r'scratch_',body)))
PY
