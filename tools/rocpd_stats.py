"""Summarise a rocprofv3 sqlite result (rocpd) as per-kernel totals: name, calls, total ms, avg us, percent.
usage: python tools/rocpd_stats.py results.db [out.csv] [steps] [skip-first-step]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    out = sys.argv[2] if len(sys.argv) > 2 else None
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    skip = len(sys.argv) > 4 and sys.argv[4] == "skip-first-step"
    if skip and steps > 1:
        # drop every kernel's launches of the first (warm-up) step: first-touch page faults and one-off packs make
        # single launches there 10x slower than in every later step
        per = {}
        for n, st, en in db.execute("select name, start, end from kernels order by start"):
            per.setdefault(n, []).append(en - st)
        rows = []
        for n, d in per.items():
            d = d[int(len(d) / steps + 0.5):] or d
            rows.append((n, len(d), sum(d), sum(d) / len(d), min(d), max(d)))
        rows.sort(key=lambda r: -r[2])
        steps -= 1
    else:
        rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                          "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["name,calls,total_ms,avg_us,min_us,max_us,percent,ms_per_step"]
    for n, c, t, a, mn, mx in rows:
        n = re.sub(r"\s+", " ", n).replace(",", ";")
        lines.append(f"\"{n[:160]}\",{c},{t / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * t / tot:.2f},{t / 1e6 / steps:.3f}")
    lines.append(f"\"TOTAL\",{sum(r[1] for r in rows)},{tot / 1e6:.3f},,,,100.0,{tot / 1e6 / steps:.3f}")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print("\n".join(l[:200] for l in lines[:45]))


if __name__ == "__main__":
    main()
