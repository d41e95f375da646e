#!/bin/bash
# smoke + GPU tests + bench sweep on the GPU box, each under its own timeout.  Usage: scripts/gpu_check.sh <tag>
TAG=${1:-check}
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || { echo "smoke failed/hung"; exit 1; }
timeout -k 10 400 python -m pytest tests -q -x -m gpu > gpurun_out/${TAG}_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/${TAG}_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 bash scripts/bench_all.sh ${TAG}_bench 2>&1 | tail -7
