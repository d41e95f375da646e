#!/bin/bash
# threads per group A/B with the region-event timing: config 4 at 256 / 512 threads, config 3 / 5 at 256
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for spec in "4 32768 512" "4 32768 256" "3 65536 256" "5 65536 256"; do set -- $spec
python bench.py --config $1 --envs $2 --threads $3 --steps 1000 --warmup 500 --min-seconds 0.1 --no-cpu --no-sweep --no-configs --no-closed-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg$1 threads $3', 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'])"
done; done
