#!/bin/bash
# start / end of every group within a launch (scripts/span.py); needs libdockauv_stamps.so.  args: config:envs:queued ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so
[ $# -eq 0 ] && set -- 3:65536:1 3:65536:32 4:32768:32 5:65536:32 2:4096:32
for spec in "$@"; do
  IFS=: read c n q x <<< "$spec"
  timeout -k 10 200 python scripts/span.py --config $c --envs $n --queued $q $x 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee gpurun_out/r3/span.txt
