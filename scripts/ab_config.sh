# one config: base library against the variant library named by $3 (same box, three rounds).  usage: ab_config.sh <config> <envs> <variant lib>
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in libdockauv.so $3; do
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib python bench.py --config $1 --envs $2 --steps 1000 --warmup 1500 --min-seconds 0.1 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'cfg$1', 'kernel_us=%.2f'%r['kernel_us'], 'us_step=%.2f'%(d['ms_per_step']*1e3))"
done; done
