#!/bin/bash
# round-2 job 2: A/B of exp / schedule variants on C3 (TB 2 vs 1, 6 vs 5 waves)
set -u
O=gpurun_out/r02_job2; mkdir -p $O
one() { # tag env...
  tag=$1; shift
  cp tools/ab_$tag.so gaussian_process_mpc_amd/csrc/libgpmpc_hip.so
  for c in ${CFGS:-C3}; do
    env "$@" python bench.py --config $c --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag $* $c', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
  done
}
for rep in 1 2; do
  one rint X=0
  one v2sb X=0
  one v2sb GPMPC_PAIR_TB=1
  one v2sb5 X=0
  one rint GPMPC_PAIR_TB=1
done 2>&1 | tee $O/ab.txt
