#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
L=gaussian_process_mpc_amd/csrc
python -m pytest tests -x -q -m gpu > $O/pytest5.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest5.log
tail -8 $O/pytest5.log
python tools/lib_ab.py --variants base=$L/libgpmpc_hip.so,GPMPC_PERSIST=0 p16=$L/libgpmpc_hip.so,GPMPC_PERSIST=16 i1=$L/libgpmpc_hip_i1.so,GPMPC_PERSIST=16 i4=$L/libgpmpc_hip_i4.so,GPMPC_PERSIST=16 w8=$L/libgpmpc_hip.so,GPMPC_PERSIST=8 \
   --shapes 300:2:1:10:256,300:4:1:10:256,512:3:1:20:256,300:4:1:10:512,200:2:1:10:1024,400:3:2:15:256,640:4:1:10:256,300:4:1:10:192,300:4:1:10:128,300:4:1:10:384,1024:4:1:20:256 > $O/ab5_persist.txt 2>&1
cat $O/ab5_persist.txt
GPMPC_LIB_PATH=$PWD/$L/libgpmpc_hip_pst.so GPMPC_PERSIST=16 python tools/persist_stamps.py 300:4:1:10:256 > $O/persist_stamps5.txt 2>&1
cat $O/persist_stamps5.txt
python tools/autotune_probe.py --verbose 300:4:1:10:256 2048:4:1:20:16 > $O/autotune5.txt 2>&1
python tools/autotune_probe.py --graph 2048:4:1:20:1 2048:4:1:20:4 1024:4:1:20:16 512:3:1:20:8 300:2:1:10:64 4096:6:1:30:1 >> $O/autotune5.txt 2>&1
cat $O/autotune5.txt
