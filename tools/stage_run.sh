#!/bin/bash
# Launch a GPU job from a FROZEN copy of the tree: gpurun snapshots /root/repo only when it has a box (minutes after the call
# starts), so a job launched from the live tree sees whatever edits happened meanwhile.  The copy lives in .stage/<name>/ (git-ignored,
# travels with the snapshot), its gpurun_out is a link to the repo's, and the job runs inside it.
#   tools/stage_run.sh <name> <timeout-seconds> <command ...>        (returns at once; log: gpurun_out/<name>.log)
set -eu
cd "$(dirname "$0")/.."
NAME=$1; TMO=$2; shift 2
rm -rf .stage/$NAME
mkdir -p .stage/$NAME gpurun_out
for d in gaussian_process_mpc_amd oracle tests tools include profiles examples bench.py __graft_entry__.py BASELINE.json *.md; do
    [ -e $d ] && cp -a $d .stage/$NAME/
done
rm -rf .stage/$NAME/gaussian_process_mpc_amd/csrc/build* .stage/$NAME/tools/ubench
rm -f .stage/$NAME/profiles/*/sanitizer_cpu.log          # (CPU sanitizer logs quote compiler flags that gpurun refuses to carry to a GPU box)
find .stage/$NAME -name __pycache__ -prune -exec rm -rf {} + 2>/dev/null || true
ln -s ../../gpurun_out .stage/$NAME/gpurun_out
ln -s gaussian_process_mpc_amd ".stage/$NAME/gaussian-process-mpc_amd" 2>/dev/null || true
# only the newest stage travels: drop the others (finished jobs)
# (only when gpurun reports NO call queued or in flight: a job launched earlier still needs its own stage)
if ! /usr/local/graft/bin/gpurun --status 2>/dev/null | grep -Eq '"in_flight": *[1-9]'; then
    for d in .stage/*; do [ "$d" != ".stage/$NAME" ] && rm -rf "$d"; done
fi
CMD="cd .stage/$NAME && $*"
# (exit code 3 = no box or slot free, nothing ran and nothing was charged: ask again after two minutes, a few times; any other outcome is final)
( for try in 1 2 3 4 5 6; do
    timeout $((TMO + 1500)) /usr/local/graft/bin/gpurun --timeout $TMO -- "$CMD" > gpurun_out/$NAME.log 2>&1; rc=$?
    grep -q "status=transient" gpurun_out/$NAME.log || break          # (nothing ran, nothing was charged: ask again)
    sleep 120
  done & )
echo "launched $NAME: $CMD"
