#!/bin/bash
# small-N mid-size batches: which kernel shape wins (fused one-launch step / staged two-kernel / scalar-broadcast 256x64)
mkdir -p gpurun_out/r02_job22
S=100:2:2:10,300:2:1:10,300:4:1:10,512:3:1:20,1024:4:1:20
B=4,8,16,32,64,128,256
for v in "default" "GPMPC_PAIR_SB=1 GPMPC_TILING=2" "GPMPC_FUSED=0 GPMPC_PAIR_SB=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  echo "== $v"
  if [ "$v" = default ]; then timeout -k 10 400 python tools/batch_map.py --quick --shapes $S --batches $B > gpurun_out/r02_job22/$tag.txt 2>&1 || exit 1
  else env $v timeout -k 10 400 python tools/batch_map.py --quick --shapes $S --batches $B > gpurun_out/r02_job22/$tag.txt 2>&1 || exit 1; fi
  grep -v amdgpu.ids gpurun_out/r02_job22/$tag.txt
done
