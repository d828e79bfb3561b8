"""Longest individual launches of a trace: python tools/rocpd_top.py results.db [n] [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = [c for c in cols if "grid" in c.lower() or "workgroup" in c.lower()]
rows = db.execute(f"select name, end-start, {', '.join(gx)} from kernels").fetchall()
# group launches of equal (name, grid) -- the same layer in every step
agg = {}
for r in rows:
    k = (re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r[0])[:70], tuple(r[2:]))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += r[1]
tot = sum(a[1] for a in agg.values())
for (name, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"{t / 1e6 / steps:7.3f} ms/step  {c / steps:5.1f} x {t / c / 1e3:8.1f} us  grid {g[0] // max(g[3], 1) if len(g) > 3 else g}  {name}")
print(f"total {tot / 1e6 / steps:.2f} ms/step")
