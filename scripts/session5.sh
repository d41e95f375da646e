#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/s5_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/s5_tests.log
bash scripts/ab_libs.sh libdockauv.so libdockauv_memonly.so 2>&1
DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so python scripts/stamps.py --config 2 2>/dev/null
bash scripts/bench_all.sh s5_bench 2>&1 | tail -7
