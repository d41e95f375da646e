#!/bin/bash
# VGPR / SGPR / scratch / occupancy / LDS / instruction mix of every f32 step-kernel instantiation (cross-compiles, no GPU)
#   bash scripts/kernel_usage.sh [extra -D flags] > profiles/r3/kernel_usage.txt
cd "$(dirname "$0")/../gym_dockauv_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -ffp-contract=on "$@" -S --cuda-device-only dockauv_kernels_f32.hip -o /tmp/k32.s -Rpass-analysis=kernel-resource-usage 2> /tmp/k32.usage
python3 - <<'PY'
import re, subprocess
from collections import Counter
txt = open('/tmp/k32.s').read(); usage = open('/tmp/k32.usage').read()
VK = {0: "BlueROV2", 1: "denseB", 2: "LAUV", 3: "mixed"}
print("step_kernel<float, vehicle, SYM, RAYS, 64, threads, LOG, TERM, WB>: instructions of the whole kernel (all roles), registers, scratch bytes/lane, waves/SIMD, SGPR spills")
rows = []
for nm in re.findall(r'\n(_ZN7dockauv11step_kernel\w+):', txt):
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
    m = re.search(r"step_kernel<float, (\d), (\w+), (\w+), 64, (\d+), (\w+), (\w+), (\w+)>", dem)
    vk, sym, rays, nt, log, term, wb = (int(m.group(1)), m.group(2) == "true", m.group(3) == "true", int(m.group(4)), m.group(5) == "true",
                                        m.group(6) == "true", m.group(7) == "true")
    body = txt.split('\n' + nm + ':')[1].split('.Lfunc_end')[0]
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith((';', '.'))]
    c = Counter('valu' if x.startswith('v_') else 'salu' if x.startswith('s_') else 'vmem' if x.startswith(('global_', 'buffer_', 'flat_')) else 'lds' if x.startswith('ds_') else 'scratch' if x.startswith('scratch_') else 'other' for x in ins)
    u = re.search(re.escape(nm) + r".*?SGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", usage, re.S)
    sg, vg, scr, occ, ssp, vsp = u.groups() if u else ("?",) * 6
    rows.append((not sym, log, vk, not rays, -nt, f"{VK[vk]:9s} sym={int(sym)} rays={int(rays)} threads={nt:3d} {'full   ' if log else ('prod+term' if term else ('prod+wb' if wb else 'product'))} | instr {len(ins):5d} valu {c['valu']:4d} salu {c['salu']:4d} vmem {c['vmem']:3d} lds {c['lds']:3d} scratch {c['scratch']:3d} | vgpr {vg:>3s} sgpr {sg:>3s} scratch {scr:>3s} B occ {occ} sgpr-spill {ssp:>3s} vgpr-spill {vsp:>3s}"))
for r in sorted(rows):
    print(r[-1])
PY
