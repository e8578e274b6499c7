#!/bin/bash
# A/B: dispatch interleave R of the scalar-broadcast pair kernel (GPMPC_RGROUP)
for c in "$@"; do
for r in 1 2 4 8 1 4; do
  GPMPC_RGROUP=$r python bench.py --config $c --no-cpu-baseline --steps 3 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c R $r', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
done
done
