#!/bin/bash
# round-2 job 1: op-cost ubench, A/B of the exp forms on C3/C4/C5, GPU tests with the new exp
set -u
O=gpurun_out/r02_job1; mkdir -p $O
tools/ubench/valu_op_cost > $O/ubench_valu_op_cost.txt 2>&1
cat $O/ubench_valu_op_cost.txt
AB_CONFIGS="C3 C4 C5" bash tools/ab_so.sh rint v2 2>&1 | tee $O/ab_exp.txt
cp tools/ab_v2.so gaussian_process_mpc_amd/csrc/libgpmpc_hip.so
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt
