#!/bin/bash
# round 4, job 4: full GPU suite on the current tree; whole-horizon kernel after the first timeline (preloads, range table, rotating priority)
set -u
O=gpurun_out/r04; mkdir -p $O
L=gaussian_process_mpc_amd/csrc
python -m pytest tests -x -q -m gpu > $O/pytest4.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest4.log
tail -15 $O/pytest4.log
python tools/lib_ab.py --variants base=$L/libgpmpc_hip.so,GPMPC_PERSIST=0 rot=$L/libgpmpc_hip.so,GPMPC_PERSIST=16 norot=$L/libgpmpc_hip_np.so,GPMPC_PERSIST=16 w8=$L/libgpmpc_hip.so,GPMPC_PERSIST=8 \
   --shapes 300:2:1:10:256,300:4:1:10:256,512:3:1:20:256,300:4:1:10:512,200:2:1:10:1024,400:3:2:15:256,640:4:1:10:256,300:4:1:10:192,300:4:1:10:128 > $O/ab4_persist.txt 2>&1
cat $O/ab4_persist.txt
GPMPC_LIB_PATH=$PWD/$L/libgpmpc_hip_pst.so GPMPC_PERSIST=16 python tools/persist_stamps.py 300:4:1:10:256 > $O/persist_stamps4.txt 2>&1
cat $O/persist_stamps4.txt
python bench.py --no-cpu-baseline > $O/bench4.json 2> $O/bench4.err; echo "bench rc $?"; tail -3 $O/bench4.err
python -c "
import json
d=json.loads([l for l in open('$O/bench4.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'])
for k,v in d.get('extras',{}).items(): print(k, v.get('value'), v.get('unit'), (v.get('roofline') or {}).get('bound'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('avg_launch_ms'), v.get('error'), round(v.get('leg_wall_s',0),1))
"
