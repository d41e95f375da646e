#!/bin/bash
# extra bench lines: config 4 at 8 x its per-GPU batch, config 5 interleaved vs vehicle-sorted
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['config']['workload'][:44], 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'GB/s=%.0f'%r['achieved'], 'frac=%.4f'%r['frac'], 'value=%.3e'%d['value'])"; }
python bench.py --config 4 --envs 262144 --steps 100 --warmup 10 --no-cpu --no-sweep 2>/dev/null | line
python bench.py --config 5 --steps 300 --warmup 30 --no-cpu --no-sweep 2>/dev/null | line
DOCKAUV_CONFIG5_SORTED=1 python bench.py --config 5 --steps 300 --warmup 30 --no-cpu --no-sweep 2>/dev/null | line
python bench.py --config 3 --envs 524288 --steps 100 --warmup 10 --no-cpu --no-sweep 2>/dev/null | line
