#!/bin/bash
# A/B two (or more) builds of the library: tools/ab_<tag>.so; configs from AB_CONFIGS (default "C3 C4")
for rep in 1 2; do
for tag in "$@"; do
  cp tools/ab_$tag.so gaussian_process_mpc_amd/csrc/libgpmpc_hip.so
  for c in ${AB_CONFIGS:-C3 C4}; do
  python bench.py --config $c --no-cpu-baseline --steps 4 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag $c', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
  done
done
done
