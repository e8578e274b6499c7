#!/bin/bash
# head / pair split of small batches of a large training set (kernel trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02_job28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "4096:6:1:30 1" "4096:6:1:30 4" "2048:4:1:20 4" "2048:4:1:20 1"; do
  set -- $cfg; tag=$(echo $1_$2 | tr ':' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 $R/tools/batch_map.py --quick --shapes $1 --batches $2 > $O/map_$tag.txt 2> $O/log_$tag.txt || { echo "trace $tag failed"; tail -3 $O/log_$tag.txt; exit 1; }
  find $O/kt_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
  rm -rf $O/kt_$tag
  echo "== $cfg"; grep "B=" $O/map_$tag.txt
  python3 - $O/kernel_stats_$tag.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in ("gpmpc", "k_roll", "k_step"))]
for r in rows[:5]:
    print("   %-64s calls %6s avg %9.2f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
