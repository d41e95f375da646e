#!/bin/bash
# parity suite + kernel times of the current library for the BASELINE configs (one line each)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_p2p.py > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -1 gpurun_out/quick_tests.log
for rep in 1 2; do
for cfg in "2 4096" "3 65536" "4 32768" "5 65536"; do
  set -- $cfg
  python bench.py --config $1 --envs $2 --steps 300 --warmup 30 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg$1', 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'value=%.3e'%d['value'])"
done; done
