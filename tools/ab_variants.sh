#!/bin/bash
# A/B of the pair-kernel variants on the GPU box: parity tests per variant, then C3 bench lines.
for v in -1 0 1; do
  echo "== GPMPC_PAIR_VARIANT=$v"
  GPMPC_PAIR_VARIANT=$v python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
  GPMPC_PAIR_VARIANT=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rollouts/s', round(d['value'],1), 'pair ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))"
done
