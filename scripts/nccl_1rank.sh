#!/bin/bash
# exercise bench.py's torch.distributed (RCCL) path with one rank on the single GPU of the box: float32 rows (+ the bf16
# sub-measurement) and --gather-dtype bf16
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for extra in "" "--gather-dtype bf16"; do
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 DOCKAUV_FORCE_DIST=1 \
  timeout -k 10 300 python bench.py --gpus 1 --steps 500 --warmup 50 --no-cpu --no-sweep $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %.3e us/step %.2f kernel_us %.2f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_us']), c['gather_dtype'], c['gather_bytes_per_rank_per_step'], c['obs_finite'], c['done_last_step_rank0'], 'alone', d['same_workload_without_gather']['ms_per_step']*1e3, 'bf16', (d.get('bf16_gather') or {}).get('ms_per_step'))"
done | tee gpurun_out/r3/nccl_1rank.txt
