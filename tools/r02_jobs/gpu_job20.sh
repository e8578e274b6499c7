#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for b in 1 2 4 8 16; do
python bench.py --config C3 --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 B=$b', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
done
for b in 1 2; do
python bench.py --config C4 --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-extras --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 B=$b', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
GPMPC_PAIR_SB=0 python bench.py --config C4 --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-extras --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 B=$b PAIR_SB=0 (two-kernel staged)', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
done
