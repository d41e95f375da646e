cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in unset 0 1; do
for cfg in "2 4096" "3 65536"; do
  set -- $cfg
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  python bench.py --config $1 --envs $2 --steps 1000 --warmup 1500 --min-seconds 0.1 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('HIP_FORCE_DEV_KERNARG=$v', 'cfg$1', 'kernel_us=%.2f'%r['kernel_us'], 'us_step=%.2f'%(d['ms_per_step']*1e3))"
done; done; done
