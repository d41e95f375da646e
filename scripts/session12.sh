#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for th in 64 128; do
  python3 bench.py --threads $th --no-cpu --no-sweep 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('unprofiled threads=$th kernel_us=%.2f ms/step=%.4f'%(d['roofline']['kernel_us'], d['ms_per_step']))"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_$th -- python3 bench.py --threads $th --no-cpu --no-sweep > gpurun_out/tr_$th.json 2>/dev/null
  f=$(find gpurun_out/tr_$th -name "*kernel_stats.csv" | head -1); sed -n 2p $f | cut -c1-160
done
