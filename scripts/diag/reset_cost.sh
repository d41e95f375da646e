cd $GRAFT_REPO_ROOT
for mode in device none device none; do
for cfg in "3 65536" "4 32768" "5 65536"; do set -- $cfg
DOCKAUV_BENCH_RESET_MODE=$mode python bench.py --config $1 --envs $2 --steps 1000 --warmup 300 --min-seconds 0.1 --no-cpu --no-sweep --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$mode', 'cfg$1', 'kernel_us=%.2f'%r['kernel_us'], 'us_step=%.2f'%(d['ms_per_step']*1e3), 'done_last', d['config']['done_last_step_rank0'])"
done; done
