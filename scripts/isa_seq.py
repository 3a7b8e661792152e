#!/usr/bin/env python3
"""Instruction-class sequence of the innermost MFMA loop of a kernel: isa_seq.py file.s <symbol substring>
   M mfma, B buffer/global load, G global store, D ds_read, W ds_write, v other vector ALU, s scalar, w s_waitcnt, X s_barrier"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and sym in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            a, b = labels[tgt], i
            nm = sum(1 for x in range(a, b) if "v_mfma" in body[x])
            if nm: loops.append((b - a, a, b, nm))
loops.sort()
_, a, b, nm = loops[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
out = []
for l in body[a:b]:
    m = re.match(r"\s+(\w+)", l)
    if not m: continue
    op = m.group(1)
    c = ("M" if "mfma" in op else "B" if op.startswith(("buffer_load", "global_load")) else "G" if op.startswith(("global_store", "buffer_store")) else
         "D" if op.startswith("ds_read") else "W" if op.startswith("ds_write") else "w" if op == "s_waitcnt" else "X" if op == "s_barrier" else
         "v" if op.startswith("v_") else "s")
    out.append(c)
print(f"loop [{a},{b}] mfma {nm}")
print("".join(out))
