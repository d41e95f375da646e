#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/s1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s1_tests.log
bash scripts/bench_all.sh s1_bench 2>&1 | tail -8
bash scripts/ab_libs.sh libdockauv.so libdockauv_hw.so 2>&1
bash scripts/pmc_sq.sh c2_4096 --config 2 2>&1 | tail -12
bash scripts/pmc_sq.sh c2_1M --config 2 --envs 1048576 2>&1 | tail -12
bash scripts/pmc_sq.sh c3 --config 3 2>&1 | tail -12
