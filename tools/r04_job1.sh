#!/bin/bash
# round 4, job 1: A/B of the pipelined weight stream against the round-3 build
set -u
O=gpurun_out/r04; mkdir -p $O
L=gaussian_process_mpc_amd/csrc
python tools/lib_ab.py --bitwise --variants base=$L/libgpmpc_hip_r03.so,GPMPC_LIB_ALLOW_MISSING=1 new=$L/libgpmpc_hip.so \
   --shapes 2048:4:1:20:1,2048:4:1:20:2,2048:4:1:20:4,4096:6:1:30:1,4096:6:1:30:2,4096:4:1:20:1,4096:4:1:20:2,1024:4:1:20:1,1024:4:1:20:4,1024:4:1:20:16,3072:4:1:20:1,512:3:1:20:1,512:3:1:20:8,2048:3:1:20:1 > $O/ab1.txt 2>&1
cat $O/ab1.txt
