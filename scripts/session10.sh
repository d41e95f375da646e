#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || { echo "smoke failed/hung"; exit 1; }
timeout -k 10 300 python -m pytest tests -q -x -m gpu > gpurun_out/s10_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/s10_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 bash scripts/bench_all.sh s10_bench 2>&1 | tail -7
DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so timeout -k 10 120 python scripts/stamps.py --config 2 2>/dev/null
