#!/bin/bash
# PMC passes for the pair kernel on the GPU box (run through gpurun from the repo root):
#   bash tools/prof_pmc.sh <tag> [bench args...]
# Each pass is its own rocprofv3 run (counters only; no tracing domains), CSV under gpurun_out/pmc_<tag>/.
set -u
TAG=$1; shift
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"
P3="FETCH_SIZE TCC_HIT_sum"
P4="WRITE_SIZE TCC_MISS_sum"
P5="SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $ROOTD/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
python3 $ROOTD/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# raw per-dispatch CSVs are large (every rocsolver dispatch of the pack build): keep the summary only
rm -rf $OUT/p[0-9]
