# config 5 only: base library against the variant library named by $1 (same box, three rounds)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in libdockauv.so $1; do
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib python bench.py --config 5 --envs 65536 --steps 1000 --warmup 1500 --min-seconds 0.1 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'cfg5', 'kernel_us=%.2f'%r['kernel_us'], 'us_step=%.2f'%(d['ms_per_step']*1e3))"
done; done
