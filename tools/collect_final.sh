#!/bin/bash
# Collects the judged artefacts of a build into gpurun_out/final/ (run through gpurun from the repo root):
# default bench line (with CPU baseline), one bench line per BASELINE config, kernel-trace stats, PMC summary.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
for c in C1 C2 C4 C5; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_final_$c.json 2>/dev/null || exit 1; done
python bench.py --config C1 --steps 20 --warmup 5 --no-cpu-baseline --graph > $O/bench_final_C1_graph.json 2>/dev/null
python bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline --graph > $O/bench_final_C2_graph.json 2>/dev/null
python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --forward-only > $O/bench_final_C3_fwd.json 2>/dev/null
for b in 1 4 16; do python bench.py --config C3 --batch $b --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_final_C3_B$b.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1 || exit 1
cd $R
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c3_final.csv \;
rm -rf $O/kt
bash tools/prof_pmc.sh final > /dev/null 2>&1
cp $R/gpurun_out/pmc_final/summary.txt $O/pmc_c3_final.txt
for f in $O/bench_final*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$(basename $f)', round(d['value'],1), d['unit'], 'ms/step', round(d['ms_per_step'],3), 'pair ms', round(r['avg_launch_ms'],4) if r['avg_launch_ms']==r['avg_launch_ms'] else None, 'frac', round(r['frac'],3) if r['frac']==r['frac'] else None)"; done
head -3 $O/kernel_stats_c3_final.csv | cut -c1-140
grep "FETCH_SIZE\|WRITE_SIZE\|SQ_INSTS_VALU \|GRBM" $O/pmc_c3_final.txt
