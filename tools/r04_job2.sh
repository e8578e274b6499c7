#!/bin/bash
# round 4, job 2: which part of the pipelined weight stream costs time; per-workgroup timeline at C3 sizes, B = 1; first run of the
# whole-horizon kernel
set -u
O=gpurun_out/r04; mkdir -p $O
L=gaussian_process_mpc_amd/csrc
python tools/lib_ab.py --bitwise --variants base=$L/libgpmpc_hip_r03.so,GPMPC_LIB_ALLOW_MISSING=1 v0=$L/libgpmpc_hip_v0.so v1=$L/libgpmpc_hip_v1.so v2=$L/libgpmpc_hip_v2.so new=$L/libgpmpc_hip.so v4=$L/libgpmpc_hip_v4.so v5=$L/libgpmpc_hip_v5.so \
   --shapes 2048:4:1:20:1,2048:4:1:20:2,4096:6:1:30:1,4096:4:1:20:1,1024:4:1:20:4,1024:4:1:20:16,2048:3:1:20:1 > $O/ab2.txt 2>&1
cat $O/ab2.txt
for v in st5b st5; do
  echo "== $v" >> $O/stamps2.txt
  GPMPC_STAMP_D=5 GPMPC_LIB_PATH=$PWD/$L/libgpmpc_hip_$v.so python tools/fused_stamps.py 2048:4:1:20:1 >> $O/stamps2.txt 2>&1
done
cat $O/stamps2.txt
python tools/env_ab.py --var GPMPC_PERSIST --values 0,16,8 300:2:1:10:256 300:4:1:10:256 512:3:1:20:256 300:4:1:10:512 200:2:1:10:1024 > $O/persist1.txt 2>&1
cat $O/persist1.txt
