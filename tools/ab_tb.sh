#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rollouts/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'pair ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))"; }
run A=1
run GPMPC_NO_XCD_SORT=1
run A=1
run GPMPC_NO_XCD_SORT=1
