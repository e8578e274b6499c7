#!/bin/bash
set -u
O=gpurun_out/r02_final; mkdir -p $O
bash tools/prof_pmc.sh C4 > $O/pmc_C4.log 2>&1; cp gpurun_out/pmc_C4/pmc_C4.txt gpurun_out/pmc_C4/pmc_C4.json $O/
bash tools/prof_pmc.sh C3 > $O/pmc_C3.log 2>&1; cp gpurun_out/pmc_C3/pmc_C3.txt gpurun_out/pmc_C3/pmc_C3.json $O/
bash tools/prof_pmc.sh C5 > $O/pmc_C5.log 2>&1; cp gpurun_out/pmc_C5/pmc_C5.txt gpurun_out/pmc_C5/pmc_C5.json $O/
for c in C3 C4 C5; do bash tools/prof_trace.sh $c > $O/trace_$c.log 2>&1; cp gpurun_out/trace_$c/kernel_stats_$c.csv gpurun_out/trace_$c/bench_traced_$c.json $O/ 2>/dev/null; done
mkdir -p profiles/r02 && cp $O/pmc_C3.json $O/pmc_C4.json $O/pmc_C5.json profiles/r02/
python bench.py --config C4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C4.json 2>/dev/null
python bench.py --gpus 2 --oversubscribe --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_one_card.json 2>/dev/null
grep -E "FETCH|WRITE|GRBM" $O/pmc_C4.txt; head -c 300 $O/bench_2rank_gloo_one_card.json; echo; python -c "
import json; d=json.load(open('$O/bench_C4.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])"
