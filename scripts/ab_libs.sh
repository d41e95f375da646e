#!/bin/bash
# A/B library variants on the same box: for each lib, kernel time at config 2 (N = 4096 / 65536 / 1M) and config 3
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  for cfg in "2 4096" "2 65536" "2 1048576" "4 32768"; do
    set -- $cfg
    DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib python bench.py --config $1 --envs $2 --steps 300 --warmup 30 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'cfg$1', 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'value=%.3e'%d['value'])"
  done
done
