#!/bin/bash
# row-chunked head kernel: kernel trace at N = 4096 / 2048, small batches, chunks off (GPMPC_HEAD_CHUNKS=1) and auto
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02_job29; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for hc in 1 auto; do
for cfg in "4096:6:1:30 1" "4096:6:1:30 4" "2048:4:1:20 4"; do
  set -- $cfg; tag=$(echo $1_$2_$hc | tr ':' '_')
  if [ $hc = auto ]; then unset GPMPC_HEAD_CHUNKS; else export GPMPC_HEAD_CHUNKS=$hc; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 $R/tools/batch_map.py --quick --shapes $1 --batches $2 > $O/map_$tag.txt 2> $O/log_$tag.txt || { echo "trace $tag failed"; tail -3 $O/log_$tag.txt; exit 1; }
  find $O/kt_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
  rm -rf $O/kt_$tag
  echo "== $cfg chunks=$hc"; grep "B=" $O/map_$tag.txt
  python3 - $O/kernel_stats_$tag.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in ("gpmpc_pair_kernel_sb<", "k_roll"))]
for r in rows[:4]:
    print("   %-64s calls %6s avg %9.2f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done
