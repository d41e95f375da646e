#!/bin/bash
# run bench.py for configs 2..5 (+ a large-N sweep of config 2) and collect the JSON lines in gpurun_out/<tag>.jsonl
TAG=${1:-bench}; shift
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG.jsonl; : > $OUT
python bench.py --steps 2000 --warmup 200 --no-cpu --no-sweep "$@" >> $OUT 2>> gpurun_out/$TAG.err
for c in 3 4 5; do python bench.py --config $c --steps 300 --warmup 30 --no-cpu --no-sweep "$@" >> $OUT 2>> gpurun_out/$TAG.err; done
for n in 65536 1048576; do python bench.py --config 2 --envs $n --steps 200 --warmup 20 --no-cpu --no-sweep "$@" >> $OUT 2>> gpurun_out/$TAG.err; done
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l); r=d["roofline"]
    print(f'{d["config"]["workload"][:8]} N={d["config"]["envs_per_gpu"]:8d} value={d["value"]:.3e} ms/step={d["ms_per_step"]:.4f} kernel_us={r["kernel_us"]:.1f} GB/s={r["achieved"]:.0f} frac={r["frac"]:.4f}')
PY
