#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench configuration (run through gpurun from the repo root):
#   bash tools/prof_trace.sh <config>     -> gpurun_out/trace_<config>/kernel_stats_<config>.csv (+ the bench line of the traced run)
set -u
CFG=$1; shift
ROOTD=$(pwd)      # the tree the job runs in (a staged copy under .stage/ when launched by tools/stage_run.sh)
OUT=$ROOTD/gpurun_out/trace_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOTD/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-legs "$@" > $OUT/bench_traced_$CFG.json 2> $OUT/trace.log || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$CFG.csv \;
rm -rf $OUT/kt
head -4 $OUT/kernel_stats_$CFG.csv | cut -c1-160
