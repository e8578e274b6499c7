#!/bin/bash
# where a mid-size batch spends its step: kernel trace of N = 1024 / 2048 at B = 16 and of N = 300 at B = 64 / 256
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02_job24; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "1024:4:1:20 16" "2048:4:1:20 16" "300:2:1:10 64" "300:2:1:10 256"; do
  set -- $cfg; tag=$(echo $1_$2 | tr ':' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 $R/tools/batch_map.py --quick --shapes $1 --batches $2 > $O/map_$tag.txt 2> $O/log_$tag.txt || { echo "trace $tag failed"; tail -3 $O/log_$tag.txt; exit 1; }
  find $O/kt_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
  rm -rf $O/kt_$tag
  echo "== $cfg"; grep "B=" $O/map_$tag.txt
  python3 - $O/kernel_stats_$tag.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rocsolver" not in r["Name"] and "Cijk" not in r["Name"]]
for r in rows[:6]:
    print("   %-64s calls %6s avg %9.2f us  total %9.2f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
