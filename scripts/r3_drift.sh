#!/bin/bash
# free-running float32 drift report for the base library and any variant libraries given as arguments
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for lib in libdockauv.so "$@"; do
  echo "== $lib"
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$lib timeout -k 10 600 python -m tests.drift_report 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3/drift.txt
