"""Average the rocprofv3 --pmc csv counters per kernel name: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "igemm"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_vgpr"] = [float(r["VGPR_Count"])]
            acc[k]["_agpr"] = [float(r["Accum_VGPR_Count"])]
            acc[k]["_lds"] = [float(r["LDS_Block_Size"])]
for k, cs in acc.items():
    print(k)
    for c in sorted(cs):
        v = cs[c]
        print(f"   {c:32s} {sum(v) / len(v):16.0f}   (n={len(v)})")
