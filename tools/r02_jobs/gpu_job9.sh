#!/bin/bash
set -u
O=gpurun_out/r02_job9; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
for f in -1 0; do
for c in C1 C2; do
  GPMPC_FUSED=$f python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSED=$f $c graph', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
  GPMPC_FUSED=$f python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSED=$f $c eager', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
done
for b in 1 4 16; do
  GPMPC_FUSED=$f python bench.py --config C3 --batch $b --steps 20 --warmup 5 --no-cpu-baseline --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSED=$f C3 B=$b graph', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
done
done 2>&1 | tee $O/latency.txt
python tools/callback_latency.py 2>/dev/null | tail -3 | tee -a $O/latency.txt
