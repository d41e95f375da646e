#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/s2_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s2_tests.log
DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_hw.so python -m pytest tests -q -m gpu > gpurun_out/s2_tests_hw.log 2>&1; echo "hw tests rc=$?"; tail -15 gpurun_out/s2_tests_hw.log
bash scripts/ab_libs.sh libdockauv.so libdockauv_hw.so libdockauv_slp.so 2>&1
bash scripts/pmc_sq.sh c2_4096b --config 2 2>&1 | tail -10
