"""Per-launch durations of the kernels whose name contains a substring, in launch order:
python tools/rocpd_launches.py results.db substring [max]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = [c for c in cols if "grid" in c.lower()]
sel = ", ".join(["name", "start", "end"] + gx)
rows = db.execute(f"select {sel} from kernels where name like ? order by start", (f"%{sys.argv[2]}%",)).fetchall()
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(r[0][:50].replace("(anonymous namespace)::", ""), f"{(r[2] - r[1]) / 1e3:9.1f} us", dict(zip(gx, r[3:])))
