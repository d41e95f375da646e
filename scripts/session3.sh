#!/bin/bash
cd $GRAFT_REPO_ROOT
for kv in 0 1; do
echo "HIP_FORCE_DEV_KERNARG=$kv"
HIP_FORCE_DEV_KERNARG=$kv python bench.py --config 2 --steps 500 --warmup 50 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'ms/step=%.5f'%d['ms_per_step'])"
done
