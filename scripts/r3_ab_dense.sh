#!/bin/bash
# same-box A/B incl. the ray-dense operating points: bench.py sub-results for the base library and variant libraries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for rep in 1 2; do
for lib in libdockauv.so "$@"; do
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib timeout -k 10 300 python bench.py --no-sweep --no-cpu --steps 1000 --warmup 500 --min-seconds 0.1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', ' '.join('%s%s=%.2f' % (s['workload'][:7], '-dense' if 'ray-dense' in s['workload'] else ('-sorted' if s.get('layout') == 'vehicle_sorted' else ''), s['kernel_us']) for s in d['configs']))"
done; done | tee gpurun_out/r3/ab_dense.txt
