#!/bin/bash
# GPU box: p2p gather tests, pace of the gather chain, then bench.py's distributed path with ONE rank (RCCL, p2p)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_p2p.py -x -q -m gpu -s > gpurun_out/p2p_tests.log 2>&1 || { tail -40 gpurun_out/p2p_tests.log; exit 1; }
grep " us \| passed\| failed" gpurun_out/p2p_tests.log
timeout -k 10 200 python scripts/p2p_pace.py 4096 > gpurun_out/p2p_pace.txt 2>&1 || { tail -20 gpurun_out/p2p_pace.txt; exit 1; }
timeout -k 10 200 python scripts/p2p_pace.py 32768 >> gpurun_out/p2p_pace.txt 2>&1 || { tail -20 gpurun_out/p2p_pace.txt; exit 1; }
grep envs gpurun_out/p2p_pace.txt
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 DOCKAUV_FORCE_DIST=1
rm -f gpurun_out/p2p_bench_1rank.jsonl gpurun_out/p2p_bench_1rank.err
for g in rccl p2p; do
  for ov in "" "--no-overlap"; do
    MASTER_PORT=295$((RANDOM % 90 + 10)) timeout -k 10 200 python bench.py --gpus 1 --steps 2000 --warmup 100 --no-cpu --no-sweep --gather $g $ov \
      >> gpurun_out/p2p_bench_1rank.jsonl 2>> gpurun_out/p2p_bench_1rank.err || { tail -20 gpurun_out/p2p_bench_1rank.err; exit 1; }
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/p2p_bench_1rank.jsonl"):
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    print(round(d["ms_per_step"] * 1e3, 2), "us/step", d["config"]["collective"][:40], "|", d["config"]["collective"][-60:])
PY
