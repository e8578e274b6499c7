#!/bin/bash
# tail-effect probe: rollouts/s and pair ms per launch as a function of the batch size (rounds of resident workgroups)
for b in 240 250 256 266 276 284 320; do
python bench.py --config C3 --batch $b --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B=$b', round(d['value'],1), 'ms/launch', round(r['avg_launch_ms'],4), 'us per trajectory-launch', round(1e3*r['avg_launch_ms']/$b,3))"
done
