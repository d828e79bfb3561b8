"""How much of a steady-state step is kernel time and how much the gaps between kernels: from a rocprofv3 sqlite
result (rocpd) of a bench run, over the kernels between the fractions `lo` and `hi` of the trace (the timed hipGraph
replays of bench.py sit in the middle: eager warm-up before them, two eager measuring steps after them).
usage: python tools/step_gaps.py results.db [lo=0.45] [hi=0.7] [top kernels to list=0]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
    k = db.execute("select start, end, name from kernels order by start").fetchall()
    k = k[int(len(k) * lo):int(len(k) * hi)]
    span = k[-1][1] - k[0][0]
    busy, cur_s, cur_e = 0, k[0][0], k[0][1]
    gaps = []
    for s, e, _ in k[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.sort()
    small = [g for g in gaps if g < 50000]
    print(f"kernels {len(k)}, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({100.0 * busy / span:.1f} %), "
          f"gaps {len(gaps)}: median {gaps[len(gaps) // 2] / 1e3:.2f} us, sum of gaps < 50 us {sum(small) / 1e6:.3f} ms, "
          f"sum of longer gaps {(sum(gaps) - sum(small)) / 1e6:.3f} ms")
    per = {}
    for s, e, n in k:
        c = per.setdefault(n, [0, 0])
        c[0] += 1
        c[1] += e - s
    tot = sum(c[1] for c in per.values())
    for n, c in sorted(per.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 0]:
        short = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:90]
        print(f"{100.0 * c[1] / tot:6.2f} %  {c[0]:5d} launches  {c[1] / c[0] / 1e3:8.1f} us  {short}")


if __name__ == "__main__":
    main()
