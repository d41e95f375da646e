#!/bin/bash
# GPU box: p2p tests (incl. riding gathers), pace, full parity suite, kernel A/B sanity (default bench line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_p2p.py -x -q -m gpu -s > gpurun_out/p2p_tests.log 2>&1 || { tail -40 gpurun_out/p2p_tests.log; exit 1; }
grep "rank 0.* us \| passed\| failed" gpurun_out/p2p_tests.log
timeout -k 10 200 python scripts/p2p_pace.py 4096 > gpurun_out/p2p_pace.txt 2>&1 || { tail -20 gpurun_out/p2p_pace.txt; exit 1; }
timeout -k 10 200 python scripts/p2p_pace.py 32768 >> gpurun_out/p2p_pace.txt 2>&1 || { tail -20 gpurun_out/p2p_pace.txt; exit 1; }
grep envs gpurun_out/p2p_pace.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_p2p.py > gpurun_out/ride_tests.log 2>&1 || { tail -30 gpurun_out/ride_tests.log; exit 1; }
tail -1 gpurun_out/ride_tests.log
for cfg in "2 4096" "2 65536" "3 65536" "4 32768"; do
  set -- $cfg
  python bench.py --config $1 --envs $2 --steps 300 --warmup 30 --no-cpu --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg$1', 'N=%8d'%d['config']['envs_per_gpu'], 'kernel_us=%.2f'%r['kernel_us'], 'frac=%.4f'%r['frac'], 'value=%.3e'%d['value'])"
done
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
        print("NON-JSON LINE ON STDOUT:", l[:80]); continue
    d = json.loads(l)
    print(round(d["ms_per_step"] * 1e3, 2), "us/step", d["config"]["collective"][:200])
PY
