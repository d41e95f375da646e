#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/s4_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s4_tests.log
bash scripts/ab_libs.sh libdockauv.so libdockauv_hw.so 2>&1
DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so python scripts/stamps.py --config 2 2>/dev/null
bash scripts/pmc_sq.sh c2_4096c --config 2 2>&1 | tail -10
