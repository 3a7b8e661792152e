#!/usr/bin/env python3
"""Where do the scratch (spill) accesses of a kernel sit relative to its MFMA loop?  usage: isa_loops.py file.s <symbol substring>"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and sym in l and l.rstrip().endswith((":", ")")) or (sym in l and re.match(r"^_Z\S+:", l)))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
mf = [i for i, l in enumerate(body) if "v_mfma" in l]
scr = [i for i, l in enumerate(body) if "scratch_" in l]
print("instructions", len(body), "mfma", len(mf), "first/last mfma", mf[0], mf[-1])
# basic blocks with back-edges
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            a, b = labels[tgt], i
            nm = sum(1 for x in mf if a <= x <= b); ns = sum(1 for x in scr if a <= x <= b)
            nv = sum(1 for x in range(a, b) if re.match(r"\s+v_(?!mfma)", body[x]))
            print(f"loop [{a},{b}] len {b-a}: mfma {nm} scratch {ns} valu(non-mfma) {nv} ds {sum(1 for x in range(a,b) if 'ds_' in body[x])} global {sum(1 for x in range(a,b) if 'global_' in body[x])} barrier {sum(1 for x in range(a,b) if 's_barrier' in body[x])}")
print("scratch ops at", scr)
