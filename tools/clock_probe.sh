#!/bin/bash
# Samples the shader clock and socket power while bench.py keeps the pair kernel busy (run through gpurun).
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/clock_probe.txt
mkdir -p $(dirname $OUT); : > $OUT
python bench.py --no-cpu-baseline --steps ${1:-300} --warmup 2 "${@:2}" > /tmp/bench_probe.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/.*: //' | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 0.3
done
wait $BP
python -c "import json; d=json.load(open('/tmp/bench_probe.json')); print('bench', d['value'], d['roofline']['avg_launch_ms'])" >> $OUT
sort $OUT | uniq -c | sort -k1,1nr | head -40
