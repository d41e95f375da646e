#!/bin/bash
# round 4 same-box A/B: the default bench line's sub-results (per-launch kernel us, resident us per step) for the tree's library
# and variant libraries (scripts/build_variant.py): bash scripts/r4_ab.sh libdockauv_x.so ...   -> gpurun_out/r4/ab_$TAG.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
TAG=${TAG:-ab}
REPS=${REPS:-2}
for rep in $(seq $REPS); do
for lib in libdockauv.so "$@"; do
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib timeout -k 10 300 python bench.py --no-sweep --no-cpu --no-closed-loop --steps 1000 --warmup 500 --min-seconds 0.1 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
def nm(s): return s['workload'][:7] + ('-dense' if 'ray-dense' in s['workload'] else ('-sorted' if s.get('layout') == 'vehicle_sorted' else ''))
print('%-28s' % '$lib', ' '.join('%s=%.2f' % (nm(s), s['kernel_us']) for s in d['configs']))
print('%-28s' % '$lib resident', ' '.join('%s=%.2f' % (nm(s), s['sequence_resident']['us_per_step_events']) for s in d['configs'] if s.get('sequence_resident')))" || echo "$lib FAILED"
done; done | tee gpurun_out/r4/ab_$TAG.txt
