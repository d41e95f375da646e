#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/micro/issue_rate.hip -o /tmp/issue_rate > gpurun_out/issue_rate.txt 2>&1
timeout -k 10 120 /tmp/issue_rate >> gpurun_out/issue_rate.txt 2>&1; echo "run rc=$?" >> gpurun_out/issue_rate.txt
cat gpurun_out/issue_rate.txt
rm -f gpurun_out/r2c3_stamps.txt
for c in "2 4096" "3 65536" "4 32768" "5 65536"; do set -- $c
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so timeout -k 10 120 python scripts/stamps.py --config $1 --envs $2 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|nanmedian" >> gpurun_out/r2c3_stamps.txt
done; cat gpurun_out/r2c3_stamps.txt
