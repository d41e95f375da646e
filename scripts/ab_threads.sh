#!/bin/bash
# threads per group (waves per 64 envs) A/B per workload; usage: ab_threads.sh
cd $GRAFT_REPO_ROOT
run() {  # config envs threads...
  local c=$1 n=$2; shift 2
  for th in "$@"; do
    python bench.py --config $c --envs $n --threads $th --steps 400 --warmup 40 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg$c threads=$th', 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'us_step=%.2f'%(d['ms_per_step']*1e3), 'value=%.3e'%d['value'])"
  done
}
run 2 4096 0 128 256
run 2 65536 0 64 128 256
run 2 262144 0 64 128
run 3 65536 0 256 512
run 4 32768 0 256 512
run 5 65536 0 256 512
