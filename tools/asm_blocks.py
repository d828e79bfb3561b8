"""Instruction mix per basic block of one kernel in a hipcc -S listing: python tools/asm_blocks.py file.s kernel-substring [min-insts]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = None
for i, l in enumerate(s):
    if re.match(r"^_Z\w*:", l) and sub in l:
        start = i
        break
assert start is not None, "kernel not found"
blocks = [["entry", []]]
for l in s[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append([t, []])
    elif t and not t.startswith(";") and not t.startswith("."):
        blocks[-1][1].append(t)
for name, ins in blocks:
    if len(ins) < mn:
        continue
    c = Counter()
    for t in ins:
        i = t.split()[0]
        if i.startswith("v_mfma"): c["mfma"] += 1
        elif i.startswith("ds_"): c["ds"] += 1
        elif i.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        elif i.startswith("v_"): c["valu"] += 1
        elif i.startswith("s_waitcnt"): c["wait"] += 1
        elif i.startswith("s_barrier"): c["barrier"] += 1
        elif i.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    print(name, len(ins), dict(c))
