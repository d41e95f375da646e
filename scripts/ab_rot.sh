#!/bin/bash
# A/B of a library variant: parity tests with the variant, then kernel times of base and variant on the same box
cd $GRAFT_REPO_ROOT
V=${1:-rot}
DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_$V.so timeout -k 10 500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_p2p.py > gpurun_out/ab_${V}_tests.log 2>&1 || { tail -30 gpurun_out/ab_${V}_tests.log; exit 1; }
tail -1 gpurun_out/ab_${V}_tests.log
for lib in libdockauv.so libdockauv_$V.so libdockauv.so libdockauv_$V.so; do
  for cfg in "2 4096" "2 65536" "3 65536" "4 32768" "5 65536"; do
    set -- $cfg
    DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib python bench.py --config $1 --envs $2 --steps 300 --warmup 30 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'cfg$1', 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'value=%.3e'%d['value'])"
  done
done | tee gpurun_out/ab_$V.txt
