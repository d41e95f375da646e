#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m tests.drift_report 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_drift.txt
bash scripts/r2_quick.sh
