#!/bin/bash
# PMC passes for the pair kernel on the GPU box (run through gpurun from the repo root):
#   [PMC_BATCH=<B of --batch, for the sidecar> PMC_STEPS=<timed steps>] bash tools/prof_pmc.sh <config> [extra bench args...]
# Each pass is its own rocprofv3 run (counters only; no tracing domains).  Writes gpurun_out/pmc_<config>/pmc_<config>.txt
# (per-launch averages) and pmc_<config>.json (the sidecar bench.py reads roofline.traffic from once it is copied to
# profiles/rNN/).  HEAD comes from tools/.githead (written on the build host before gpurun: the box has no .git).
set -u
CFG=$1; shift
ROOTD=$(pwd)      # the tree the job runs in (a staged copy under .stage/ when launched by tools/stage_run.sh)
OUT=$ROOTD/gpurun_out/pmc_$CFG
mkdir -p $OUT
HEAD=$(cat $ROOTD/tools/.githead 2>/dev/null || echo unknown)
BATCH=${PMC_BATCH:-$(python3 -c "import sys; sys.path.insert(0,'$ROOTD'); from gaussian_process_mpc_amd.synth import CONFIGS; c=CONFIGS['$CFG']; print(c['B']//8 if '$CFG'=='C4' else c['B'])")}
STEPS=${PMC_STEPS:-1}
CMD="python3 bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras --no-legs $*"
# the inverse kernel matrices come from an UNPROFILED run: rocprofv3 --pmc segfaults inside rocSOLVER's 4096^2 LU (C4)
# (PMC_KC: its own cache file for a run whose extra arguments change the training set, e.g. --n-train)
KC=${PMC_KC:-/tmp/kinv_$CFG.pt}
[ -f $KC ] || python3 $ROOTD/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-legs --kinv-cache $KC "$@" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"
P3="FETCH_SIZE TCC_HIT_sum"
P4="WRITE_SIZE TCC_MISS_sum"
P5="SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $ROOTD/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras --no-legs --kinv-cache $KC "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
{ echo "# rocprofv3 --pmc (5 separate passes) of: $CMD"; echo "# HEAD $HEAD"; python3 $ROOTD/tools/pmc_summary.py $OUT --json $OUT/pmc_$CFG.json --config $CFG --batch $BATCH --command "$CMD" --head $HEAD; } > $OUT/pmc_$CFG.txt 2>&1
grep -E "FETCH_SIZE|WRITE_SIZE|GRBM|SQ_ACTIVE_INST_VALU " $OUT/pmc_$CFG.txt
# raw per-dispatch CSVs are large (every rocsolver dispatch of the pack build): keep the summaries only
rm -rf $OUT/p[0-9]
