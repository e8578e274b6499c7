#!/bin/bash
set -u
O=gpurun_out/r02_job10; mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in C1 C2; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$c -- python3 $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
find $R/$O/kt_$c -name "*kernel_stats.csv" -exec cp {} $R/$O/kernel_stats_$c.csv \;
find $R/$O/kt_$c -name "*kernel_trace.csv" -exec cp {} $R/$O/kernel_trace_$c.csv \;
rm -rf $R/$O/kt_$c
echo "== $c"; head -4 $R/$O/kernel_stats_$c.csv | cut -c1-150
done
cd $R
python - <<'PY'
import csv
for c in ("C1","C2"):
    rows=list(csv.DictReader(open(f"gpurun_out/r02_job10/kernel_trace_{c}.csv")))
    rows=[r for r in rows if "step_fused" in r["Kernel_Name"] or "roll_tail" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    # gaps between consecutive kernels of one rollout (last 400 kernels)
    rows=rows[-420:]
    gaps=[int(rows[i+1]["Start_Timestamp"])-int(rows[i]["End_Timestamp"]) for i in range(len(rows)-1)]
    dur=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
    import statistics
    print(c, "median kernel ns", statistics.median(dur), "median gap ns", statistics.median(gaps))
PY
