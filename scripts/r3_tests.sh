#!/bin/bash
# GPU suite (without the p2p rehearsal), output incl. the per-trajectory parity lines -> gpurun_out/r3/tests.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -s --deselect tests/test_gpu_p2p.py "$@" > gpurun_out/r3/tests.log 2>&1
rc=$?
grep -a "^\[parity" gpurun_out/r3/tests.log | sort -u > gpurun_out/r3/parity_lines.txt
tail -30 gpurun_out/r3/tests.log
exit $rc
