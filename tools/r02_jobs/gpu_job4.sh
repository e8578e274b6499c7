#!/bin/bash
set -u
O=gpurun_out/r02_job4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 | tee $O/pytest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_survey_8d_slots'], r['first_step_variant'], d.get('pcie_inclusive_rollouts_per_s'), d['cpu_baseline'])"
python bench.py --gpus 2 --oversubscribe --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_one_card.json 2> $O/bench_2rank.err; echo "2-rank rc=$?"; cat $O/bench_2rank_gloo_one_card.json | cut -c1-400; tail -3 $O/bench_2rank.err
python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_2rank_refused.txt 2>&1; echo "2-rank without GPUs rc=$? (must be non-zero)"; tail -2 $O/bench_2rank_refused.txt
bash tools/prof_trace.sh C3 2>&1 | tail -6
bash tools/prof_pmc.sh C3 2>&1 | tail -8
