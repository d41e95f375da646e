#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in "2 4096" "3 65536" "4 32768"; do set -- $c
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so timeout -k 10 120 python scripts/stamps.py --config $1 --envs $2 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|nanmedian"
done
