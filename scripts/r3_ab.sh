#!/bin/bash
# same-box A/B of kernel times (event-timed, bench.py) for the base library and variant libraries: configs 2-5, three rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
LIBS="libdockauv.so $@"
for rep in 1 2 3; do
for lib in $LIBS; do
for cfg in "2 4096" "3 65536" "4 32768" "5 65536"; do
  set -- $cfg
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib timeout -k 10 120 python bench.py --config $1 --envs $2 --steps 1000 --warmup 1500 --min-seconds 0.1 --no-cpu --no-sweep --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'cfg$1', 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'us_step=%.2f'%(d['ms_per_step']*1e3))"
done; done; done | tee gpurun_out/r3/ab.txt
