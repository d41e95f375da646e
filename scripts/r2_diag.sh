#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== product lib, auto threads"; timeout -k 10 120 python scripts/diag/general_vs_sym.py 0 2>&1 | grep -v amdgpu.ids
echo "== product lib, 256 threads"; timeout -k 10 120 python scripts/diag/general_vs_sym.py 256 2>&1 | grep -v amdgpu.ids
echo "== product lib, no rays (SimpleCurrent)"; timeout -k 10 120 python scripts/diag/general_vs_sym.py 0 SimpleCurrentDocking3d 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  echo "== variant $v, auto threads"; DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$v timeout -k 10 120 python scripts/diag/general_vs_sym.py 0 2>&1 | grep -v amdgpu.ids
done
