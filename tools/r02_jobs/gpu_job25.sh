#!/bin/bash
# workgroups per CU of the scalar-broadcast kernel (GPMPC_SB_OCC): never capped (-1), chosen per launch (unset), fixed 4..7
for occ in -1 auto 4 5 6 7; do
  echo "== GPMPC_SB_OCC=$occ"
  if [ $occ = auto ]; then unset GPMPC_SB_OCC; else export GPMPC_SB_OCC=$occ; fi
  timeout -k 10 300 python tools/batch_map.py --quick --shapes 1024:4:1:20 --batches 16,24,32,64 2>&1 | grep "B="
  timeout -k 10 300 python tools/batch_map.py --quick --shapes 2048:4:1:20 --batches 8,16,32,48 2>&1 | grep "B="
  timeout -k 10 300 python tools/batch_map.py --quick --shapes 300:2:1:10 --batches 256 2>&1 | grep "B="
  timeout -k 10 300 python tools/batch_map.py --quick --shapes 4096:6:1:30 --batches 2,4 2>&1 | grep "B="
done
