#!/bin/bash
# bench.py --gpus 2 in its DEFAULT transport (one all-gather per step), two rank processes on the box's one GPU with gloo
# standing in for RCCL (RCCL refuses two ranks on one device): exercises the N > 1 code path incl. --gather-dtype bf16
cd $GRAFT_REPO_ROOT
export WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=297$((RANDOM % 90 + 10)) DOCKAUV_DIST_BACKEND=gloo
for mode in "" "--gather-dtype bf16"; do
  pids=()
  for r in 0 1; do
    RANK=$r timeout -k 10 300 python bench.py --gpus 2 --steps 100 --warmup 10 --min-seconds 0.01 --no-cpu --no-sweep $mode > gpurun_out/g2_r$r.out 2> gpurun_out/g2_r$r.err &
    pids+=($!)
  done
  rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
  if [ $rc -ne 0 ]; then tail -15 gpurun_out/g2_r*.err; exit 1; fi
  python - <<PY
import json
d = json.loads(open("gpurun_out/g2_r0.out").read()); c = d["config"]
print("[$mode]", "n_gpus", d["n_gpus"], "value %.3e" % d["value"], "us/step %.1f" % (d["ms_per_step"] * 1e3), c["gather_dtype"], c["gather_bytes_per_rank_per_step"], c["obs_finite"], c["done_last_step_rank0"], c["backend"], "alone us/step %.2f" % (d["same_workload_without_gather"]["ms_per_step"] * 1e3), c["collective"][:60])
PY
  export MASTER_PORT=$((MASTER_PORT + 1))
done
