#!/bin/bash
AB_CONFIGS="C3 C4 C5" bash tools/ab_so.sh base rowsum 2>&1 | grep -v amdgpu
cp tools/ab_rowsum.so gaussian_process_mpc_amd/csrc/libgpmpc_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
