#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_storage.py tests/test_gpu_reset.py -x -q -m gpu 2>&1 | tail -40
