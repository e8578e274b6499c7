#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs for the pair kernels: per-launch averages of every counter, as text (stdout) and, with
--json OUT, as the sidecar bench.py reads `roofline.traffic` from (profiles/rNN/pmc_<config>.json).

    pmc_summary.py <dir with p*/ pass directories> [--json OUT --config C3 --batch 256 --command "..." --head <sha>]
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--json")
ap.add_argument("--config", default="")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--command", default="")
ap.add_argument("--head", default="")
ap.add_argument("--forward-only", action="store_true")
a = ap.parse_args()

acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(a.root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if not any(t in k for t in ("pair_kernel", "k_step_fused", "k_traj_persist")):
            continue
        e = acc[(k.split("(")[0].replace("void ", "").strip(), r["Counter_Name"])]
        e[0] += float(r["Counter_Value"]); e[1] += 1
kernels = defaultdict(dict)
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{k:50s} {c:28s} per-launch avg {s / n:18.1f}  (n={n})")
    kernels[k][c] = s / n
    kernels[k].setdefault("_launches", {})[c] = n
if a.json:
    # dominant kernel = the one the chip spends most time in: launches x GRBM_GUI_ACTIVE (falls back to launch count)
    def weight(k):
        v = kernels[k]
        n = max(v["_launches"].values())
        return n * v.get("GRBM_GUI_ACTIVE", 1.0)
    dom = max(kernels, key=weight) if kernels else None
    out = {"head": a.head, "command": a.command, "config": a.config, "batch_per_gpu": a.batch, "want_grad": not a.forward_only, "shared_lambda": "--shared-lambda" in a.command,
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch (rocprofv3 derived counters); other counters raw, per launch",
           "kernels": {k: {c: v for c, v in d.items() if c != "_launches"} for k, d in kernels.items()},
           "launches_profiled": {k: max(d["_launches"].values()) for k, d in kernels.items()}}
    if dom:
        out["dominant_kernel"] = dict(name=dom, **{c: v for c, v in kernels[dom].items() if c != "_launches"})
    with open(a.json, "w") as f:
        json.dump(out, f, indent=1)
