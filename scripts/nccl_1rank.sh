#!/bin/bash
# exercise bench.py's torch.distributed (RCCL) path with one rank on the single GPU of the box
cd $GRAFT_REPO_ROOT
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 DOCKAUV_FORCE_DIST=1 \
  python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu --no-sweep
