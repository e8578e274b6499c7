#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest7.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest7.log
tail -12 $O/pytest7.log
for v in "--cl-newton" "" ; do python bench.py --closed-loop $v > "$O/closed_loop7$(echo $v | tr -d ' ').json" 2>$O/cl7.err || tail -5 $O/cl7.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/closed_loop7*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'median', round(d['value'],3), 'mean', round(d['step_ms']['mean'],3), 'max', round(d['step_ms']['max'],2), 'split', {k:round(v,3) for k,v in d['split_ms_mean'].items()}, 'excl solve', {k:(round(v,3) if v else v) for k,v in d['step_ms_excluding_solve'].items()}, 'inv', d['inverse_update_ms'])
PY
python tools/autotune_probe.py 200:2:1:10:64 200:2:1:10:128 200:4:1:10:64 200:4:1:10:96 300:4:1:10:64 300:4:1:10:96 300:4:1:10:128 300:4:1:10:160 300:2:1:10:128 300:2:1:10:160 400:4:1:10:64 400:4:1:10:128 400:3:1:10:128 448:4:1:10:96 512:4:1:10:64 512:4:1:10:128 > $O/autotune7.txt 2>&1
python tools/autotune_probe.py --graph 200:2:1:10:32 200:4:1:10:32 300:4:1:10:48 300:4:1:10:64 300:4:1:10:96 400:4:1:10:32 400:4:1:10:64 512:4:1:10:32 >> $O/autotune7.txt 2>&1
cat $O/autotune7.txt
