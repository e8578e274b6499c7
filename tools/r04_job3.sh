#!/bin/bash
# round 4, job 3: pipelined weight stream with counted waits (unconditional reload); timeline at C4 sizes, B = 1; whole-horizon kernel timeline;
# shared-lambda kernel with the weights one column ahead
set -u
O=gpurun_out/r04; mkdir -p $O
L=gaussian_process_mpc_amd/csrc
python tools/lib_ab.py --bitwise --variants base=$L/libgpmpc_hip_r03.so,GPMPC_LIB_ALLOW_MISSING=1 p0=$L/libgpmpc_hip_p0.so p1=$L/libgpmpc_hip_p1.so p2=$L/libgpmpc_hip_p2.so p3=$L/libgpmpc_hip_p3.so \
   --shapes 4096:6:1:30:1,4096:6:1:30:2,4096:4:1:20:1,3072:4:1:20:1,2048:4:1:20:1,1024:4:1:20:16 > $O/ab3.txt 2>&1
cat $O/ab3.txt
GPMPC_STAMP_D=7 GPMPC_LIB_PATH=$PWD/$L/libgpmpc_hip_st7.so python tools/fused_stamps.py 4096:6:1:30:1 > $O/stamps3_c4b1.txt 2>&1
cat $O/stamps3_c4b1.txt
GPMPC_LIB_PATH=$PWD/$L/libgpmpc_hip_pst.so GPMPC_PERSIST=16 python tools/persist_stamps.py 300:4:1:10:256 > $O/persist_stamps.txt 2>&1
GPMPC_LIB_PATH=$PWD/$L/libgpmpc_hip_pst.so GPMPC_PERSIST=16 python tools/persist_stamps.py 1024:4:1:10:256 >> $O/persist_stamps.txt 2>&1
cat $O/persist_stamps.txt
python tools/lib_ab.py --shared-lambda --variants nopre=$L/libgpmpc_hip_p0.so pre=$L/libgpmpc_hip_p1.so --shapes 2048:4:1:20:256,2048:4:1:20:64,1024:4:1:20:256,4096:6:1:30:32,512:3:1:20:256 > $O/ab3_shared.txt 2>&1
cat $O/ab3_shared.txt
