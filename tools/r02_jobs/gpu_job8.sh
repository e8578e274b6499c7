#!/bin/bash
# B = 1 solver-callback path: kernel durations (trace) and end-to-end latency
set -u
O=gpurun_out/r02_job8; mkdir -p $O
R=$(pwd)
for c in C1 C2; do
  python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c graph', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
  python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c eager', round(d['value'],1), 'rollouts/s', round(d['ms_per_step'],4), 'ms')"
done
cd /tmp && export TMPDIR=/tmp
for c in C1 C2; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$c -- python3 $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
find $R/$O/kt_$c -name "*kernel_stats.csv" -exec cp {} $R/$O/kernel_stats_$c.csv \;
rm -rf $R/$O/kt_$c
echo "== $c"; head -6 $R/$O/kernel_stats_$c.csv | cut -c1-150
done
cd $R
python tools/callback_latency.py 2>/dev/null | tail -5
