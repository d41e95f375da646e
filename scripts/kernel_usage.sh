#!/bin/bash
# print VGPR / scratch / occupancy / instruction counts of the f32 step kernels
cd "$(dirname "$0")/../gym_dockauv_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -S --cuda-device-only dockauv_kernels_f32.hip -o /tmp/k32.s -Rpass-analysis=kernel-resource-usage 2> /tmp/k32.usage
python3 - <<'PY'
import re
from collections import Counter
txt=open('/tmp/k32.s').read(); usage=open('/tmp/k32.usage').read()
for nm in re.findall(r'\n(_ZN7dockauv11step_kernel\w+):', txt):
    body=txt.split('\n'+nm+':')[1].split('.Lfunc_end')[0]
    ins=[l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith((';','.'))]
    c=Counter('valu' if x.startswith('v_') else 'salu' if x.startswith('s_') else 'vmem' if x.startswith(('global_','buffer_','flat_')) else 'lds' if x.startswith('ds_') else 'scr' if x.startswith('scratch_') else 'o' for x in ins)
    m=re.search(re.escape(nm)+r".*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+)", usage, re.S)
    print(nm[24:50], "instr", len(ins), dict(c), "vgpr/scratch/occ/sspill", m.groups() if m else None)
PY
