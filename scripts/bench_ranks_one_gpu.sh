#!/bin/bash
# Rehearsal of `bench.py --gpus N` with N rank processes that all use the box's ONE GPU (gloo instead of RCCL for the
# reference all-gather and the scalar reductions; the p2p transport is the real one).  Usage: bench_ranks_one_gpu.sh [N]
cd $GRAFT_REPO_ROOT
N=${1:-2}
export WORLD_SIZE=$N LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=296$((RANDOM % 90 + 10)) DOCKAUV_DIST_BACKEND=gloo
for mode in "--gather auto" "--gather auto --no-overlap"; do
  pids=()
  for r in $(seq 0 $((N - 1))); do
    RANK=$r timeout -k 10 300 python bench.py --gpus $N --steps ${STEPS:-1000} --warmup ${WARMUP:-100} --no-cpu --no-sweep $mode \
      > gpurun_out/ranks_${N}_r$r.out 2> gpurun_out/ranks_${N}_r$r.err &
    pids+=($!)
  done
  rc=0
  for p in "${pids[@]}"; do wait $p || rc=1; done
  if [ $rc -ne 0 ]; then tail -20 gpurun_out/ranks_${N}_r*.err; exit 1; fi
  python - <<PY
import json
d = json.loads(open("gpurun_out/ranks_${N}_r0.out").read())
print("$mode:", "n_gpus", d["n_gpus"], "value %.3e" % d["value"], "us/step %.2f" % (d["ms_per_step"] * 1e3), "|", d["config"]["collective"][:150], d["config"].get("gather_note", ""))
PY
  export MASTER_PORT=$((MASTER_PORT + 1))
done
