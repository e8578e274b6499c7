#!/usr/bin/env python3
"""From a `rocprofv3 --kernel-trace` CSV: how busy the GPU is over the rollout kernels of the LAST part of the run -- the span
from the first to the last gpmpc kernel of the final `--last` fraction of dispatches, the fraction of that span covered by at
least one running kernel, the average number of kernels running concurrently, and time per kernel class.
    python tools/trace_overlap.py <dir with *_kernel_trace.csv> [--last 0.3]"""
import argparse, csv, glob, os
ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--last", type=float, default=0.3)
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gpmpc" in n or "k_roll" in n or "k_step_fused" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("<")[0].replace("void ", "")))
rows.sort()
rows = rows[int(len(rows) * (1.0 - a.last)):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows])
busy = conc = 0
depth, last = 0, t0
for t, d in ev:
    if depth > 0:
        busy += t - last
        conc += depth * (t - last)
    depth += d
    last = t
per = {}
for s, e, n in rows:
    per[n] = per.get(n, 0) + (e - s)
print(f"{len(rows)} rollout kernels over {(t1 - t0) / 1e3:.1f} us: GPU busy {busy / (t1 - t0):.3f} of the span, "
      f"{conc / max(busy, 1):.2f} kernels running on average while busy; summed kernel time {sum(per.values()) / 1e3:.1f} us")
for n, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print(f"   {n:32s} {v / 1e3:10.1f} us")
