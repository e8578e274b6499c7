#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the pair kernel only: bash tools/pmc_fetch.sh <tag> [bench args...]   (env passes through)
TAG=$1; shift
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out/pmcf_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/p1 -- python3 $ROOTD/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/p1.log 2>&1 || echo "pass failed"
python3 $ROOTD/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
rm -rf $OUT/p1
