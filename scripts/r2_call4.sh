#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/r2_quick.sh || exit 1
rm -f gpurun_out/r2c4_stamps.txt
for c in "2 4096" "3 65536" "4 32768" "5 65536"; do set -- $c
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so timeout -k 10 120 python scripts/stamps.py --config $1 --envs $2 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|nanmedian" >> gpurun_out/r2c4_stamps.txt
done; cat gpurun_out/r2c4_stamps.txt
