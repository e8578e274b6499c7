#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest6.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest6.log
tail -25 $O/pytest6.log
for v in "--cl-newton" "" "--cl-distinct"; do python bench.py --closed-loop $v > "$O/closed_loop6$(echo $v | tr -d ' ').json" 2>$O/cl6.err || tail -5 $O/cl6.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/closed_loop6*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'median', round(d['value'],3), 'mean', round(d['step_ms']['mean'],3), 'max', round(d['step_ms']['max'],2), 'split', {k:round(v,3) for k,v in d['split_ms_mean'].items()}, 'excl solve', {k:(round(v,3) if v else v) for k,v in d['step_ms_excluding_solve'].items()}, 'inv', d['inverse_update_ms'])
PY
python tools/autotune_probe.py 300:4:1:10:256 300:4:1:10:128 300:4:1:10:512 300:2:1:10:256 300:2:1:10:64 512:3:1:20:256 512:3:1:20:64 1024:4:1:20:32 1024:4:1:20:64 1024:4:1:20:256 2048:4:1:20:16 2048:4:1:20:32 2048:4:1:20:64 2048:4:1:20:256 4096:6:1:30:8 4096:6:1:30:128 > $O/autotune6.txt 2>&1
python tools/autotune_probe.py --graph 300:4:1:10:1 300:4:1:10:8 300:4:1:10:32 300:2:1:10:16 512:3:1:20:1 512:3:1:20:16 1024:4:1:20:1 1024:4:1:20:2 1024:4:1:20:4 1024:4:1:20:8 1024:4:1:20:16 2048:4:1:20:1 2048:4:1:20:2 2048:4:1:20:4 2048:4:1:20:8 4096:6:1:30:1 4096:6:1:30:2 >> $O/autotune6.txt 2>&1
cat $O/autotune6.txt
