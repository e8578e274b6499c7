#!/bin/bash
for jt in 128 192 256 320; do
for c in C3 C4; do
GPMPC_JT0=$jt python bench.py --config $c --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('JT0=$jt $c', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
done
done
