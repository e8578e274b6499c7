#!/bin/bash
# small N, large batches: 256x256 tiles (1 or 2 trajectories per wave) against the default plan
S="--shapes 100:2:2:10,300:2:1:10,300:4:1:10,512:3:1:20 --batches 128,256,512,1024,2048"
echo "== default"; timeout -k 10 400 python tools/batch_map.py --quick $S 2>&1 | grep -v amdgpu
for tb in 1 2; do echo "== GPMPC_PAIR_SB=1 GPMPC_TILING=0 GPMPC_PAIR_TB=$tb"; GPMPC_PAIR_SB=1 GPMPC_TILING=0 GPMPC_PAIR_TB=$tb timeout -k 10 400 python tools/batch_map.py --quick $S 2>&1 | grep -v amdgpu; done
echo "== GPMPC_PAIR_SB=1 GPMPC_TILING=2"; GPMPC_PAIR_SB=1 GPMPC_TILING=2 timeout -k 10 400 python tools/batch_map.py --quick $S 2>&1 | grep -v amdgpu
