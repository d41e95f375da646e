#!/bin/bash
# the default bench line (+ a table of its sub-results) -> gpurun_out/r4/bench_$TAG.json / .txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
TAG=${TAG:-default}
timeout -k 10 900 python bench.py "$@" > gpurun_out/r4/bench_$TAG.json 2> gpurun_out/r4/bench_$TAG.err || { tail -20 gpurun_out/r4/bench_$TAG.err; exit 1; }
python scripts/r4_table.py gpurun_out/r4/bench_$TAG.json | tee gpurun_out/r4/bench_$TAG.txt
