#!/bin/bash
# per-launch and resident us per step for a list of "config envs threads" (bench.py --no-configs) -> gpurun_out/r4/threads_resident.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for spec in "$@"; do set -- $spec
timeout -k 10 150 python bench.py --config $1 --envs $2 --threads $3 --steps 1000 --warmup 300 --min-seconds 0.1 --no-cpu --no-sweep --no-configs --no-closed-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d.get('sequence_resident') or {}
print('cfg$1 envs %8d threads %3d' % ($2, $3), 'launch_us=%.2f' % r['kernel_us'], 'frac=%.4f' % r['frac'], 'resident_us=%.2f' % s.get('us_per_step_events', float('nan')), 'resident_frac=%.4f' % s.get('frac_of_8TBps', float('nan')))" || echo "$spec FAILED"
done | tee gpurun_out/r4/threads_resident.txt
