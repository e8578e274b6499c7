#!/bin/bash
# big (256x256) vs mid (256x64) tiling of the scalar-broadcast kernel around the crossover
mkdir -p gpurun_out/r02_job23
for t in 0 2; do
  echo "== GPMPC_PAIR_SB=1 GPMPC_TILING=$t"
  GPMPC_PAIR_SB=1 GPMPC_TILING=$t timeout -k 10 500 python tools/batch_map.py --quick --shapes 2048:4:1:20 --batches 24,32,40,48,64 2>&1 | grep -v amdgpu
  GPMPC_PAIR_SB=1 GPMPC_TILING=$t timeout -k 10 500 python tools/batch_map.py --quick --shapes 4096:6:1:30 --batches 2,4,6,8,12 2>&1 | grep -v amdgpu
  GPMPC_PAIR_SB=1 GPMPC_TILING=$t timeout -k 10 500 python tools/batch_map.py --quick --shapes 1024:4:1:20,512:3:1:20 --batches 64,96,128,192,256 2>&1 | grep -v amdgpu
done
