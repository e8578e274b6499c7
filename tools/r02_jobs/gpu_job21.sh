#!/bin/bash
# batch-size map at the default plan, then the mid-size batches with the tiling forced either way
mkdir -p gpurun_out/r02_job21
timeout -k 10 600 python tools/batch_map.py > gpurun_out/r02_job21/map_default.txt 2>&1 || exit 1
cat gpurun_out/r02_job21/map_default.txt
for t in 0 2; do
  echo "GPMPC_PAIR_SB=1 GPMPC_TILING=$t"
  GPMPC_PAIR_SB=1 GPMPC_TILING=$t timeout -k 10 300 python tools/batch_map.py --shapes 1024:4:1:20,2048:4:1:20 --batches 4,8,16,32,64 > gpurun_out/r02_job21/map_tiling$t.txt 2>&1 || exit 1
  cat gpurun_out/r02_job21/map_tiling$t.txt
done
