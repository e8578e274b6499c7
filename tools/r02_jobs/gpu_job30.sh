#!/bin/bash
# PMC of the 256x64 scalar-broadcast pair kernel on mid-size batches (where the per-step time does not shrink with B)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"
for cfg in "1024:4:1:20 16" "2048:4:1:20 8" "4096:6:1:30 1"; do
  set -- $cfg; tag=$(echo $1_$2 | tr ':' '_'); O=$R/gpurun_out/r02_job30/$tag; mkdir -p $O
  i=0
  for P in "$P1" "$P2"; do i=$((i+1)); timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 $R/tools/batch_map.py --quick --shapes $1 --batches $2 > $O/p$i.log 2>&1 || echo "pass failed"; done
  echo "== N:ds:da:H = $1, B = $2"; python3 $R/tools/pmc_summary.py $O | grep -v "true, true"
  rm -rf $O/p[0-9]
done
