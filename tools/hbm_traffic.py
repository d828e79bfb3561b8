"""Per-launch HBM traffic of the contraction kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/hbm_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ csv trees> <out.json>

The output records which kernels it was measured on: the SHA-256 over the library's kernel sources (the same digest
bench.py computes at run time and compares: a traffic figure of other kernels is reported as null) and, where a git
checkout is at hand, the commit.

rocprofv3 reports both counters in KiB-like units of 1 KB = 1024 B per dispatch.  On gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B per lane) streaming reads at 64 bytes each, so the read bytes are 2 x FETCH_SIZE;
WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.  traffic = 2 * FETCH_SIZE + WRITE_SIZE.
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
from collections import defaultdict

FAMILIES = {
    "conv_fwd": ("igemm_nt_persist_kernel<0", "igemm_nt_fast_kernel<0", "igemm_nt_bf16_kernel<0"),
    "conv_dgrad": ("igemm_nt_persist_kernel<1", "igemm_nt_persist_kernel<2", "igemm_nt_fast_kernel<1",
                   "igemm_nt_fast_kernel<2", "igemm_nt_bf16_kernel<1"),
    "conv_wgrad": ("igemm_tn_wgrad",),
    "bn": ("bn_apply_kernel", "bn_bwd_apply_kernel", "bn_reduce_kernel"),
}


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{d}/{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for fam, pats in FAMILIES.items():
                if any(p in r["Kernel_Name"] for p in pats):
                    acc[fam][0] += float(r["Counter_Value"])
                    acc[fam][1] += 1
    return acc


def kernel_sources_digest(root):
    """SHA-256 over the names and bytes of video-classification_amd/csrc/*.hip and *.h, in name order."""
    h = hashlib.sha256()
    src = os.path.join(root, "video-classification_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            with open(os.path.join(src, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def main():
    d, out = sys.argv[1], sys.argv[2]
    fetch, write = collect(d, "FETCH_SIZE"), collect(d, "WRITE_SIZE")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    res = {"kernel_sources_sha256": kernel_sources_digest(root), "commit": commit,
           "unit": "bytes per launch (average over the launches of 3 eager steps of bench.py)",
           "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests at 64 B); write bytes = WRITE_SIZE x 1024",
           "families": {}}
    for fam in FAMILIES:
        if fetch[fam][1] == 0 or write[fam][1] == 0:
            continue
        rd = 2.0 * fetch[fam][0] * 1024 / fetch[fam][1]
        wr = write[fam][0] * 1024 / write[fam][1]
        res["families"][fam] = {"launches_sampled": fetch[fam][1], "steps_sampled": 3, "read_bytes": rd,
                                "write_bytes": wr, "traffic_bytes": rd + wr,
                                "traffic_bytes_per_step": (rd + wr) * fetch[fam][1] / 3.0}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
