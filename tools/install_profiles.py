#!/usr/bin/env python3
"""Copy the artefacts of tools/collect_r02.sh (gpurun_out/r02_final/) into profiles/r02/ and rewrite the numbers quoted in
profiles/r02/README.md ("How to recompute ...") from them."""
import csv, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out", "r02_final"), os.path.join(R, "profiles", "r02")
for c in ("C3", "C4", "C5"):
    for f in (f"pmc_{c}.txt", f"pmc_{c}.json", f"kernel_stats_{c}.csv", f"bench_{c}.json", f"bench_traced_{c}.json"):
        shutil.copy(os.path.join(O, f), P)
for f in ("bench_C1", "bench_C1_graph", "bench_C2", "bench_C2_graph", "bench_C3_fwd", "bench_C3_B1", "bench_C3_B4", "bench_C3_B16"):
    shutil.copy(os.path.join(O, f + ".json"), P)
open(os.path.join(P, "bench_2rank_gloo_one_card.json"), "w").write("".join(l for l in open(os.path.join(O, "bench_2rank_gloo_one_card.json")) if l.startswith("{")))
shutil.copy(os.path.join(O, "callback_latency.txt"), P)
V = {}
for c in ("C3", "C4", "C5"):
    d = json.load(open(os.path.join(P, f"bench_{c}.json"))); r = d["roofline"]
    p = json.load(open(os.path.join(P, f"pmc_{c}.json"))); k = p["dominant_kernel"]
    cyc = k["GRBM_GUI_ACTIVE"] / 8
    top = [x for x in csv.DictReader(open(os.path.join(P, f"kernel_stats_{c}.csv"))) if k["name"] in x["Name"]][0]
    t = json.load(open(os.path.join(P, f"bench_traced_{c}.json")))["roofline"]["avg_launch_ms"]
    V[c] = dict(v=d["value"], ms=r["avg_launch_ms"], frac=r["frac"], slots=r["frac_survey_8d_slots"], ach=r["achieved"], tr=float(top["AverageNs"]) / 1e6,
                calls=top["Calls"], tev=t, valu=k["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, ghz=cyc / (float(top["AverageNs"]) * 1e-9) / 1e9,
                traffic=r["traffic"] / 1e9, pcie=d.get("pcie_inclusive_rollouts_per_s"), fwd=d.get("forward_only_rollouts_per_s"), head=p["head"][:8])
    print(c, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in V[c].items()})
cb = json.load(open(os.path.join(P, "bench_C3.json")))["cpu_baseline"]
S = {f: json.load(open(os.path.join(P, f + ".json"))) for f in ("bench_C1_graph", "bench_C2_graph", "bench_C3_B1", "bench_C3_B4", "bench_C3_B16")}
cl = open(os.path.join(P, "callback_latency.txt")).read()
cbl = re.findall(r"(C[12]): objective\+gradient callback pair ([0-9.]+) ms", cl)
p = os.path.join(P, "README.md"); s = open(p).read()
i = s.index("  its bench line is `bench_traced_<config>.json`): C3"); j = s.index("* `achieved` = pairs × flops ÷ time:")
s = s[:i] + f"""  its bench line is `bench_traced_<config>.json`): C3 `gpmpc_pair_kernel_sb<5, 2, 4, true, false>` {V['C3']['tr']:.3f} ms over {V['C3']['calls']} calls
  (incl. the warm-up) vs {V['C3']['ms']:.3f} ms by events in `bench_C3.json` and {V['C3']['tev']:.3f} ms in the traced run itself; C4 `<7, 1, 6, true, false>`
  {V['C4']['tr']:.3f} ms ({V['C4']['calls']} calls) vs {V['C4']['ms']:.3f} / {V['C4']['tev']:.3f}; C5 `gpmpc_pair_kernel_sbf<5, 4, true>` {V['C5']['tr']:.3f} ms ({V['C5']['calls']} calls) vs {V['C5']['ms']:.3f} / {V['C5']['tev']:.3f}.
""" + s[j:]
i = s.index("* `achieved` = pairs × flops ÷ time:"); j = s.index("* `traffic` =")
s = s[:i] + f"""* `achieved` = pairs × flops ÷ time: C3 {V['C3']['ach']:.1f} TFLOP/s → `frac` {V['C3']['frac']:.3f}; C4 {V['C4']['ach']:.1f} → {V['C4']['frac']:.3f}; C5 {V['C5']['ach']:.1f} → {V['C5']['frac']:.3f}.
  `frac_survey_8d_slots` (2D+23 slots vs 39.3e12/s): {V['C3']['slots']:.3f} / {V['C4']['slots']:.3f} / {V['C5']['slots']:.3f}.  (Devices of the pool differ: the same kernels gave
  0.750 / 0.721 on the slowest and 0.789 / 0.776 on the fastest device met in the round -- C3 6.08–6.41 k, C4 0.57–0.615 k rollouts/s.)
""" + s[j:]
s = re.sub(r"C3 [0-9.]+, C4 [0-9.]+, C5 [0-9.]+ at\n  [0-9.]+–[0-9.]+ GHz", f"C3 {V['C3']['valu']:.3f}, C4 {V['C4']['valu']:.2f}, C5 {V['C5']['valu']:.2f} at\n  {min(v['ghz'] for v in V.values()):.2f}–{max(v['ghz'] for v in V.values()):.2f} GHz", s)
s = re.sub(r"default bench line: [0-9.]+ k rollouts/s \(6\.08–6\.41 k across devices of the pool\), cpu_baseline \([^)]*\), PCIe-inclusive [0-9.]+ k, objective-only [0-9.]+ k",
           f"default bench line: {V['C3']['v'] / 1e3:.2f} k rollouts/s (6.08–6.41 k across devices of the pool), cpu_baseline (16 threads of an EPYC 9575F: faithful {cb['value']:.3f} rollouts/s min-of-5, 1 thread {cb['faithful_1_thread']:.3f}, O(N²) torch {cb['o2_value']:.3f}, C port {cb['c_port_value']:.1f} / {cb['c_port_value_1_thread']:.2f}), PCIe-inclusive {V['C3']['pcie'] / 1e3:.2f} k, objective-only {V['C3']['fwd'] / 1e3:.2f} k", s)
s = re.sub(r"[0-9.]+ k rollouts/s per GPU \(B = 128\), [0-9.]+ k \(full covariance\)", f"{V['C4']['v'] / 1e3:.3f} k rollouts/s per GPU (B = 128), {V['C5']['v'] / 1e3:.2f} k (full covariance)", s)
s = re.sub(r"C1 [0-9.]+ ms, C2 [0-9.]+ ms, N = 2048 B = 1 / 4 / 16 [0-9.]+ / [0-9.]+ / [0-9.]+ ms per step",
           f"C1 {S['bench_C1_graph']['ms_per_step']:.3f} ms, C2 {S['bench_C2_graph']['ms_per_step']:.3f} ms, N = 2048 B = 1 / 4 / 16 {S['bench_C3_B1']['ms_per_step']:.2f} / {S['bench_C3_B4']['ms_per_step']:.2f} / {S['bench_C3_B16']['ms_per_step']:.1f} ms per step", s)
if len(cbl) == 2:
    s = re.sub(r"Python included: C1 [0-9.]+ ms, C2 [0-9.]+ ms", f"Python included: C1 {cbl[0][1]} ms, C2 {cbl[1][1]} ms", s)
open(p, "w").write(s)
