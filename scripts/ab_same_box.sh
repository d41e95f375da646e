#!/bin/bash
# same-box A/B of two libraries, alternating: kernel_us per config.  Usage: ab_same_box.sh <variant> [reps]
cd $GRAFT_REPO_ROOT
V=${1:-pre}; REPS=${2:-3}
for rep in $(seq $REPS); do
for lib in libdockauv_$V.so libdockauv.so; do
  for cfg in "2 4096" "2 65536" "3 65536" "4 32768" "5 65536"; do
    set -- $cfg
    DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib python bench.py --config $1 --envs $2 --steps 400 --warmup 40 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'cfg$1', 'N=%d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'us_step=%.2f'%(d['ms_per_step']*1e3))"
  done
done; done | tee gpurun_out/ab_same_box.txt
python - <<'PY'
import collections, statistics
acc = collections.defaultdict(list)
for l in open("gpurun_out/ab_same_box.txt"):
    p = l.split()
    acc[(p[1], p[2], p[0])].append((float(p[3].split("=")[1]), float(p[4].split("=")[1])))
for k in sorted(acc):
    print(k, "kernel_us median %.2f" % statistics.median(x[0] for x in acc[k]), "us/step median %.2f" % statistics.median(x[1] for x in acc[k]))
PY
