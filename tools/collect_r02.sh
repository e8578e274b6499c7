#!/bin/bash
# Collects the judged artefacts of the round-2 build into gpurun_out/r02_final/ (run through gpurun from the repo root;
# tools/.githead must hold the HEAD the snapshot was taken at).  Copy the result into profiles/r02/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_final
mkdir -p $O
cd $R
for c in C3 C4 C5; do bash tools/prof_pmc.sh $c > $O/pmc_$c.log 2>&1; cp gpurun_out/pmc_$c/pmc_$c.txt gpurun_out/pmc_$c/pmc_$c.json $O/ 2>/dev/null; done
for c in C3 C4 C5; do bash tools/prof_trace.sh $c > $O/trace_$c.log 2>&1; cp gpurun_out/trace_$c/kernel_stats_$c.csv gpurun_out/trace_$c/bench_traced_$c.json $O/ 2>/dev/null; done
# the sidecars of THIS run feed roofline.traffic of the bench lines below
mkdir -p profiles/r02 && cp $O/pmc_C3.json $O/pmc_C4.json $O/pmc_C5.json profiles/r02/ 2>/dev/null
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err || echo "bench C3 failed"
for c in C4 C5; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$c.json 2>/dev/null || echo "bench $c failed"; done
for c in C1 C2; do
  python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_$c.json 2>/dev/null
  python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --graph > $O/bench_${c}_graph.json 2>/dev/null
done
python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --forward-only > $O/bench_C3_fwd.json 2>/dev/null
for b in 1 4 16; do python bench.py --config C3 --batch $b --steps 20 --warmup 5 --no-cpu-baseline --graph > $O/bench_C3_B$b.json 2>/dev/null; done
python bench.py --gpus 2 --oversubscribe --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_one_card.json 2>/dev/null
python tools/callback_latency.py 2>/dev/null | tail -2 > $O/callback_latency.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.lstrip().startswith("{")][-1]); r = d["roofline"]
g = lambda v, n=3: None if v is None or v != v else round(v, n)
print(sys.argv[1].split("/")[-1], g(d["value"], 1), d["unit"], "ms/step", g(d["ms_per_step"]), "kernel ms", g(r["avg_launch_ms"], 4),
      "frac", g(r["frac"]), "slots", g(r["frac_survey_8d_slots"]), "traffic", r["traffic"], "n_gpus", d["n_gpus"])
PY
done
cat $O/callback_latency.txt
