#!/bin/bash
# Collects the judged artefacts of the round-5 build into gpurun_out/r05c/ (run through gpurun from the repo root or a staged copy;
# tools/.githead must hold the HEAD the snapshot was taken at).  Copy the result into profiles/r05/ afterwards.
#   bash tools/collect_r05.sh [trace|pmc|pmc1|pmc2|pmc3|bench|maps|all]
set -u
R=$(pwd)
O=$R/gpurun_out/r05c
mkdir -p $O
STEP=${1:-all}
if [ $STEP = all ] || [ $STEP = trace ]; then
# kernel traces (rocprofv3 --kernel-trace --stats)
for c in C3 C4 C5; do bash tools/prof_trace.sh $c > $O/trace_$c.log 2>&1; cp gpurun_out/trace_$c/kernel_stats_$c.csv gpurun_out/trace_$c/bench_traced_$c.json $O/ 2>/dev/null; done
bash tools/prof_trace.sh C3 --shared-lambda > $O/trace_C3_shared.log 2>&1; cp gpurun_out/trace_C3/kernel_stats_C3.csv $O/kernel_stats_C3_shared.csv; cp gpurun_out/trace_C3/bench_traced_C3.json $O/bench_traced_C3_shared.json
bash tools/prof_trace.sh C5 --shared-lambda > $O/trace_C5_shared.log 2>&1; cp gpurun_out/trace_C5/kernel_stats_C5.csv $O/kernel_stats_C5_shared.csv; cp gpurun_out/trace_C5/bench_traced_C5.json $O/bench_traced_C5_shared.json
bash tools/prof_trace.sh C3 --batch 1 > $O/trace_C3_B1.log 2>&1; cp gpurun_out/trace_C3/kernel_stats_C3.csv $O/kernel_stats_C3_B1.csv; cp gpurun_out/trace_C3/bench_traced_C3.json $O/bench_traced_C3_B1.json
bash tools/prof_trace.sh C3 --batch 8 > $O/trace_C3_B8.log 2>&1; cp gpurun_out/trace_C3/kernel_stats_C3.csv $O/kernel_stats_C3_B8.csv; cp gpurun_out/trace_C3/bench_traced_C3.json $O/bench_traced_C3_B8.json
bash tools/prof_trace.sh C4 --batch 1 > $O/trace_C4_B1.log 2>&1; cp gpurun_out/trace_C4/kernel_stats_C4.csv $O/kernel_stats_C4_B1.csv; cp gpurun_out/trace_C4/bench_traced_C4.json $O/bench_traced_C4_B1.json
bash tools/prof_trace.sh C5 --batch 1 > $O/trace_C5_B1.log 2>&1; cp gpurun_out/trace_C5/kernel_stats_C5.csv $O/kernel_stats_C5_B1.csv; cp gpurun_out/trace_C5/bench_traced_C5.json $O/bench_traced_C5_B1.json
bash tools/prof_trace.sh C3 --n-train 300 --batch 256 > $O/trace_N300_B256.log 2>&1; cp gpurun_out/trace_C3/kernel_stats_C3.csv $O/kernel_stats_N300_B256.csv; cp gpurun_out/trace_C3/bench_traced_C3.json $O/bench_traced_N300_B256.json
bash tools/prof_trace.sh C3 --n-train 300 --batch 256 --shared-lambda > $O/trace_N300_B256_shared.log 2>&1; cp gpurun_out/trace_C3/kernel_stats_C3.csv $O/kernel_stats_N300_B256_shared.csv; cp gpurun_out/trace_C3/bench_traced_C3.json $O/bench_traced_N300_B256_shared.json
fi
if [ $STEP = all ] || [ $STEP = pmc ] || [ $STEP = pmc1 ]; then
# PMC (separate passes, counters only)
for c in C3 C4 C5; do bash tools/prof_pmc.sh $c > $O/pmc_$c.log 2>&1; cp gpurun_out/pmc_$c/pmc_$c.txt gpurun_out/pmc_$c/pmc_$c.json $O/ 2>/dev/null; done
bash tools/prof_pmc.sh C3 --shared-lambda > $O/pmc_C3_shared.log 2>&1; cp gpurun_out/pmc_C3/pmc_C3.txt $O/pmc_C3_shared.txt; cp gpurun_out/pmc_C3/pmc_C3.json $O/pmc_C3_shared.json
fi
if [ $STEP = all ] || [ $STEP = pmc ] || [ $STEP = pmc3 ]; then
PMC_KC=/tmp/kinv_N300.pt PMC_BATCH=256 PMC_STEPS=3 bash tools/prof_pmc.sh C3 --n-train 300 --batch 256 > $O/pmc_N300_B256.log 2>&1; cp gpurun_out/pmc_C3/pmc_C3.txt $O/pmc_N300_B256.txt; cp gpurun_out/pmc_C3/pmc_C3.json $O/pmc_N300_B256.json
PMC_KC=/tmp/kinv_N300s.pt PMC_BATCH=256 PMC_STEPS=3 bash tools/prof_pmc.sh C3 --n-train 300 --batch 256 --shared-lambda > $O/pmc_N300_B256_shared.log 2>&1; cp gpurun_out/pmc_C3/pmc_C3.txt $O/pmc_N300_B256_shared.txt; cp gpurun_out/pmc_C3/pmc_C3.json $O/pmc_N300_B256_shared.json
fi
if [ $STEP = all ] || [ $STEP = pmc ] || [ $STEP = pmc2 ]; then
PMC_BATCH=1 PMC_STEPS=3 bash tools/prof_pmc.sh C3 --batch 1 > $O/pmc_C3_B1.log 2>&1; cp gpurun_out/pmc_C3/pmc_C3.txt $O/pmc_C3_B1.txt; cp gpurun_out/pmc_C3/pmc_C3.json $O/pmc_C3_B1.json
PMC_BATCH=8 PMC_STEPS=3 bash tools/prof_pmc.sh C3 --batch 8 > $O/pmc_C3_B8.log 2>&1; cp gpurun_out/pmc_C3/pmc_C3.txt $O/pmc_C3_B8.txt; cp gpurun_out/pmc_C3/pmc_C3.json $O/pmc_C3_B8.json
PMC_BATCH=1 PMC_STEPS=2 bash tools/prof_pmc.sh C4 --batch 1 > $O/pmc_C4_B1.log 2>&1; cp gpurun_out/pmc_C4/pmc_C4.txt $O/pmc_C4_B1.txt; cp gpurun_out/pmc_C4/pmc_C4.json $O/pmc_C4_B1.json
PMC_BATCH=1 PMC_STEPS=3 bash tools/prof_pmc.sh C5 --batch 1 > $O/pmc_C5_B1.log 2>&1; cp gpurun_out/pmc_C5/pmc_C5.txt $O/pmc_C5_B1.txt; cp gpurun_out/pmc_C5/pmc_C5.json $O/pmc_C5_B1.json
fi
if [ $STEP = all ] || [ $STEP = bench ]; then
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err; grep '^BENCH_FULL ' $O/bench_C3.err | sed 's/^BENCH_FULL //' > $O/bench_C3_full.json || echo "bench C3 failed"
python bench.py --config C3 --shared-lambda --no-cpu-baseline --full-json --no-legs > $O/bench_C3_shared.json 2>/dev/null || echo "bench C3 shared failed"
python bench.py --config C5 --shared-lambda --steps 5 --warmup 2 --no-cpu-baseline --full-json > $O/bench_C5_shared.json 2>/dev/null || echo "bench C5 shared failed"
for b in 1 8 64; do python bench.py --config C5 --shared-lambda --batch $b --steps 10 --warmup 3 --no-cpu-baseline --full-json --no-extras > $O/bench_C5_shared_B$b.json 2>/dev/null; done
for c in C4 C5; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --full-json > $O/bench_$c.json 2>/dev/null || echo "bench $c failed"; done
for c in C1 C2; do
  python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --full-json > $O/bench_$c.json 2>/dev/null
  python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --full-json --graph > $O/bench_${c}_graph.json 2>/dev/null
done
python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --full-json --forward-only > $O/bench_C3_fwd.json 2>/dev/null
for b in 1 4 8 16 32; do python bench.py --config C3 --batch $b --steps 20 --warmup 5 --no-cpu-baseline --full-json --graph > $O/bench_C3_B$b.json 2>/dev/null; done
python bench.py --config C4 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --full-json --graph > $O/bench_C4_B1.json 2>/dev/null
for b in 1 2 4 8; do python bench.py --config C5 --batch $b --steps 20 --warmup 5 --no-cpu-baseline --full-json --no-extras > $O/bench_C5_B$b.json 2>/dev/null; done
python bench.py --config C3 --n-train 300 --batch 256 --shared-lambda --steps 20 --warmup 5 --no-cpu-baseline --full-json --no-extras > $O/bench_N300_B256_shared.json 2>/dev/null
for n in 300 512; do python bench.py --config C3 --n-train $n --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --full-json --no-extras > $O/bench_N${n}_B256.json 2>/dev/null; done
for b in 8 16 32; do python bench.py --config C3 --n-train 1024 --batch $b --steps 50 --warmup 10 --no-cpu-baseline --full-json --no-extras --graph > $O/bench_N1024_B$b.json 2>/dev/null; done
python bench.py --gpus 2 --oversubscribe --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --full-json > $O/bench_2rank_gloo_one_card.json 2>$O/bench_2rank.err
for v in "" "--cl-distinct" "--cl-rebuild" "--cl-newton"; do python bench.py --closed-loop $v > "$O/closed_loop$(echo $v | tr -d ' ').json" 2>/dev/null; done
python bench.py --closed-loop --cl-newton --cl-starts 16 > $O/closed_loop--cl-newton--cl-starts16.json 2>/dev/null
python tools/callback_latency.py 2>/dev/null | tail -2 > $O/callback_latency.txt
fi
if [ $STEP = all ] || [ $STEP = maps ]; then
python tools/batch_map.py > $O/batch_size_map.txt 2>&1
python tools/batch_map.py --shapes 300:2:1:10,300:4:1:10,512:3:1:20 --batches 192,256,512,1024 >> $O/batch_size_map.txt 2>&1
python tools/batch_map.py --fullcov --shapes 300:4:1:10,1024:4:1:20,2048:4:1:20 --batches 1,2,4,8,16,64 > $O/batch_size_map_fullcov.txt 2>&1
fi
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.lstrip().startswith("{")][-1]); r = d["roofline"]
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e); sys.exit(0)
g = lambda v, n=3: None if v is None or v != v else round(v, n)
print(sys.argv[1].split("/")[-1], g(d["value"], 1), d["unit"], "ms/step", g(d["ms_per_step"]), "kernel", r["kernel"].split(" ")[0], "bound", r["bound"], "launch ms", g(r["avg_launch_ms"], 4),
      "frac", g(r["frac"]), "hbm_frac", g(r.get("hbm_frac")), "issue_util", g(r.get("issue_util")), "traffic", r["traffic"], "n_gpus", d["n_gpus"])
PY
done
