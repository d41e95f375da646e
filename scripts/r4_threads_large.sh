#!/bin/bash
# round 4: threads per group beyond residency, same box, base library (lib/libdockauv_base.so when present) against the tree's.
#   args: "lib config envs threads" ... ; output -> gpurun_out/r4/threads_large.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
OUT=${OUT:-gpurun_out/r4/threads_large.txt}
if [ $# -eq 0 ]; then
  LIBS="libdockauv.so"; [ -f gym_dockauv_amd/lib/libdockauv_base.so ] && LIBS="libdockauv_base.so libdockauv.so"
  SPECS=()
  for L in $LIBS; do
    for N in 131072 262144 524288 1048576; do SPECS+=("$L 3 $N 64"); done
    for N in 262144 524288 1048576; do SPECS+=("$L 3 $N 256"); done
    SPECS+=("$L 4 1048576 64" "$L 4 1048576 256" "$L 5 1048576 64" "$L 5 1048576 256" "$L 2 1048576 64")
  done
  set -- "${SPECS[@]}"
fi
for spec in "$@"; do set -- $spec
DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$1 timeout -k 10 150 python bench.py --config $2 --envs $3 --threads $4 --steps 200 --warmup 100 --min-seconds 0.1 --no-cpu --no-sweep --no-configs --no-closed-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-20s cfg$2 envs %8d threads %3d' % ('$1', $3, $4), 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'env-steps/s=%.3g'%d['value'])" || echo "$spec FAILED"
done | tee $OUT
