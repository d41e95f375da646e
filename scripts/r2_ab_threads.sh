#!/bin/bash
# threads per group x library A/B for the obstacle configs (same box)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in libdockauv.so "$@"; do
for cfg in "3 65536" "4 32768" "5 65536"; do
  set -- $cfg
  for th in 128 256 512; do
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib python bench.py --config $1 --envs $2 --threads $th --steps 1000 --warmup 1000 --min-seconds 0.1 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'cfg$1 threads=$th', 'kernel_us=%.2f'%r['kernel_us'], 'us_step=%.2f'%(d['ms_per_step']*1e3))"
done; done; done; done | tee gpurun_out/ab_threads2.txt
