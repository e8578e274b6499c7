#!/bin/bash
# bench lines for the BASELINE configs on one GPU (no CPU baseline leg)
for c in C1 C2; do python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$c graph\", round(d[\"value\"],1), round(d[\"ms_per_step\"],3))"; done
for c in C1 C2 C3 C4 C5; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', 'rollouts/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'pair ms', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'launches', r['launches'])"
done
python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --forward-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C3 fwd-only', 'rollouts/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'pair ms', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3))"
