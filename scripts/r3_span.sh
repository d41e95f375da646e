#!/bin/bash
# start / end of every group within a launch (scripts/span.py) for the BASELINE configs; needs libdockauv_stamps.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so
for cfg in "3 65536" "4 32768" "5 65536" "2 4096" "3 262144" "3 1048576" "2 1048576"; do
  set -- $cfg
  timeout -k 10 200 python scripts/span.py --config $1 --envs $2 2>&1 | grep -v Warning
done | tee gpurun_out/r3/span.txt
