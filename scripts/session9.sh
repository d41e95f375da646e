#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || { echo "smoke failed/hung"; exit 1; }
timeout -k 10 300 python -m pytest tests -q -x -m gpu > gpurun_out/s9_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/s9_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 bash scripts/bench_all.sh s9_bench 2>&1 | tail -7
timeout -k 10 200 python bench.py > gpurun_out/s9_default_bench.json 2> gpurun_out/s9_default_bench.err; tail -c 1500 gpurun_out/s9_default_bench.json
