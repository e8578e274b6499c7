#!/usr/bin/env python3
"""Condensed view of one kernel of an ISA listing: python tools/isa_view.py <file.s> <mangled-substring> [regex]
prints, with line numbers relative to the kernel start, every line matching the regex (default: memory ops, barriers, waits, branches, spills)."""
import re, sys
s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"scratch_|buffer_load|global_load|global_store|s_barrier|s_waitcnt|s_cbranch|s_branch|^\.LBB|s_load|s_dcache|s_endpgm|ds_read|ds_write")
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start + 1, len(s)) if s[i].startswith("_Z") or ".end_amdhsa_kernel" in s[i] or s[i].startswith("\t.section"))
print(s[start].split(":")[0], "lines", end - start)
n = 0
for i in range(start, end):
    l = s[i].strip()
    if not l or l.startswith(";"):
        continue
    n += 1
    if pat.search(l):
        print(f"{i - start:6d} {l[:130]}")
