#!/usr/bin/env python3
"""Register / spill table of every kernel in one translation unit:
    python tools/kres.py <file.hip> [-DGPMPC_PAIR_D=5 ...] [--grep PATTERN]
(hipcc -Rpass-analysis=kernel-resource-usage, demangled, one line per kernel)."""
import re, subprocess, sys
args = sys.argv[1:]
pat = None
if "--grep" in args:
    i = args.index("--grep"); pat = re.compile(args[i + 1]); del args[i:i + 2]
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", args[0], "-o", "/dev/null"] + args[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: +(?:Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark: +([A-Za-z ]+(?:\[[a-zA-Z/]+\])?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    elif "error" in line:
        print(line)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("void ", "").split("(")[0]
    if pat and not pat.search(n):
        continue
    print(f"{n:58s} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', 0):3d} sgpr {r.get('TotalSGPRs', r.get('SGPRs', -1)):4d} spill v{r.get('VGPRs Spill', 0):3d} s{r.get('SGPRs Spill', 0):3d} "
          f"scratch {r.get('ScratchSize [bytes/lane]', 0):4d} occ {r.get('Occupancy [waves/SIMD]', -1)} lds {r.get('LDS Size [bytes/block]', -1)}")
