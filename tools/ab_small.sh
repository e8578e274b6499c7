#!/bin/bash
for rep in 1 2; do
for tag in "$@"; do
  cp tools/ab_$tag.so gaussian_process_mpc_amd/csrc/libgpmpc_hip.so
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --batch 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag C3 B1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
  for c in C1 C2; do python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --graph | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag $c graph', round(d['value'],1), round(d['ms_per_step'],3))"; done
done
done
