#!/bin/bash
# 256x128 tiles (GPMPC_JT0=128 on the large-tile list) against 256x64 for mid-size batches
S1="--shapes 1024:4:1:20 --batches 8,16,24,32,64"; S2="--shapes 2048:4:1:20 --batches 4,8,16,32"
echo "== 256x64 (default mid)"; for s in "$S1" "$S2"; do GPMPC_PAIR_SB=1 GPMPC_TILING=2 timeout -k 10 300 python tools/batch_map.py --quick $s 2>&1 | grep "B="; done
for tb in 1 2; do
echo "== 256x128, $tb trajectories per wave"; for s in "$S1" "$S2"; do GPMPC_JT0=128 GPMPC_PAIR_SB=1 GPMPC_TILING=0 GPMPC_PAIR_TB=$tb timeout -k 10 300 python tools/batch_map.py --quick $s 2>&1 | grep "B="; done
done
echo "== 256x256, 1 trajectory per wave"; for s in "$S1" "$S2"; do GPMPC_PAIR_SB=1 GPMPC_TILING=0 GPMPC_PAIR_TB=1 timeout -k 10 300 python tools/batch_map.py --quick $s 2>&1 | grep "B="; done
