#!/bin/bash
# scripts/p2p_soak.py with N rank processes on one GPU.  Usage: p2p_soak.sh [N] [REGIONS]
cd $GRAFT_REPO_ROOT
N=${1:-2}; REG=${2:-2000}
PORT=298$((RANDOM % 90 + 10))
pids=()
for r in $(seq 1 $((N - 1))); do
  timeout -k 10 500 python scripts/p2p_soak.py $r $N $PORT $REG > gpurun_out/soak_${N}_r$r.log 2>&1 &
  pids+=($!)
done
timeout -k 10 500 python scripts/p2p_soak.py 0 $N $PORT $REG 2>&1 | grep "rank 0" | tee gpurun_out/soak_${N}_r0.log
rc=${PIPESTATUS[0]}
for p in "${pids[@]}"; do wait $p || rc=1; done
tail -qn1 gpurun_out/soak_${N}_r*.log
exit $rc
