#!/bin/bash
# in-kernel time lines (scripts/stamps.py, diagnostic build).  args: "config envs [--dense]" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so
[ $# -eq 0 ] && set -- "2 4096" "3 65536" "4 32768" "5 65536" "3 65536 --dense" "4 32768 --dense"
for spec in "$@"; do
  set -- $spec
  timeout -k 10 200 python scripts/stamps.py --config $1 --envs $2 $3 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee gpurun_out/r4/stamps.txt
