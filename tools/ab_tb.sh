#!/bin/bash
# A/B harness for the pair-kernel shapes on the GPU box (C3): each line is one bench.py run with the given overrides.
#   GPMPC_PAIR_SB=0|1   staged (pair_kernel.h) vs scalar-broadcast (pair_kernel_sb.h / pair_kernel_sbf.h) kernels
#   GPMPC_PAIR_TB=1|2|4 trajectories per workgroup sharing an M_ij load
#   GPMPC_NO_XCD_SORT=1 plain (unit, row, column) order of the work list
run() { echo "== $*"; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rollouts/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'pair ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))"; }
run GPMPC_PAIR_SB=1
run GPMPC_PAIR_SB=1 GPMPC_PAIR_TB=2
run GPMPC_PAIR_SB=0 GPMPC_PAIR_TB=2
run GPMPC_PAIR_SB=0 GPMPC_PAIR_TB=4
run GPMPC_PAIR_SB=1 GPMPC_NO_XCD_SORT=1
