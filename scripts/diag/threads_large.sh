#!/bin/bash
# threads per group at sizes beyond residency (many rounds of groups).  args: "lib config envs threads" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
[ $# -eq 0 ] && set -- "libdockauv.so 3 1048576 256" "libdockauv.so 3 1048576 64" "libdockauv.so 3 262144 256" "libdockauv.so 3 262144 64" "libdockauv.so 4 1048576 256" "libdockauv.so 4 1048576 64"
for spec in "$@"; do set -- $spec
DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$1 timeout -k 10 120 python bench.py --config $2 --envs $3 --threads $4 --steps 200 --warmup 100 --min-seconds 0.1 --no-cpu --no-sweep --no-configs --no-closed-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 cfg$2 envs $3 threads $4', 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'env-steps/s=%.3g'%d['value'])"
done | tee gpurun_out/r3/threads_large.txt
