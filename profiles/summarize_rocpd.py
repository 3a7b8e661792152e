#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into per-kernel statistics.
usage: summarize_rocpd.py <results.db> [skip_first_n_dispatches_per_kernel]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = db.execute(f"select {name_col}, start, end, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels order by start").fetchall() \
    if {"grid_x", "lds_size"} <= set(cols) else [r + (0,) * 7 for r in db.execute(f"select {name_col}, start, end from kernels order by start")]
stat = {}
for r in rows:
    n = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", ""))
    n = re.sub(r"^void ", "", n)
    if len(r) > 3 and "conv_mfma" in n:
        n += f" grid=({r[3] // max(r[5],1)},{r[4]})"
    d = (r[2] - r[1]) / 1e3
    s = stat.setdefault(n, [0, 0.0, 1e30, 0.0, r[6:]])
    s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
tot = sum(s[1] for s in stat.values())
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}  lds/vgpr/agpr/sgpr")
for n, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:100]:100s} {s[0]:6d} {s[1]:12.1f} {s[1]/s[0]:10.2f} {s[2]:10.2f} {s[3]:10.2f} {100*s[1]/tot:6.2f}  {s[4]}")
print(f"{'TOTAL':100s} {sum(s[0] for s in stat.values()):6d} {tot:12.1f}")
