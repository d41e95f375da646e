#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "3 65536" "4 32768" "4 262144" "5 65536"; do
  set -- $cfg
  for th in 256 512; do
    python bench.py --config $1 --envs $2 --threads $th --steps 100 --warmup 10 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg$1 threads=$th', 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.1f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'value=%.3e'%d['value'])"
  done
done
