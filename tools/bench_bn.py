"""Microbenchmark of the BatchNorm kernels at ResNet-50 training shapes (B*T = 512 frames, 16 groups).
Reports GB/s of algorithmic traffic per kernel.  `--hot` times a single group-sized slice repeatedly
(working set <= 160 MB) to show what the memory-side cache gives."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_classification_amd import _lib as L


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--groups", type=int, default=16)
    ap.add_argument("--hot", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = [(112, 64), (56, 64), (56, 256), (28, 128), (28, 512), (14, 256), (14, 1024), (7, 512), (7, 2048)]
    G = a.groups
    print(f"{'shape':>12} {'MB':>7} | {'stats':>12} {'apply':>12} {'apply+res':>12} {'bwd_red':>12} {'bwd_app':>12} {'bwd_app+res':>12}   (ms, GB/s)")
    for hw, C in shapes:
        N = a.frames if not a.hot else a.frames // G
        g = G if not a.hot else 1
        M = N * hw * hw
        rpg = M // g
        y = torch.randn(M, C, device=dev).bfloat16()
        d = torch.randn(M, C, device=dev).bfloat16()
        res = torch.randn(M, C, device=dev).bfloat16()
        out = torch.empty_like(y)
        dy = torch.empty_like(y)
        dres = torch.empty_like(y)
        s = torch.zeros(2, g, C, device=dev)
        stat = torch.rand(4, g, C, device=dev) + 0.5
        S = y.numel() * 2 / 1e6
        st = L.stream()
        ks = {
            "stats": (1, lambda: L.call("vc_bn_stats", 1, L.ptr(y), L.ptr(s[0]), L.ptr(s[1]), g, rpg, C, st)),
            "apply": (2, lambda: L.call("vc_bn_apply", 1, L.ptr(y), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(None), L.ptr(out), g, rpg, C, 1, st)),
            "apply+res": (3, lambda: L.call("vc_bn_apply", 1, L.ptr(y), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(res), L.ptr(out), g, rpg, C, 1, st)),
            "bwd_red": (2, lambda: L.call("vc_bn_bwd_reduce", 1, L.ptr(y), L.ptr(d), L.ptr(None), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(s[0]), L.ptr(s[1]), g, rpg, C, 1, st)),
            "bwd_app": (3, lambda: L.call("vc_bn_bwd_apply", 1, L.ptr(y), L.ptr(d), L.ptr(None), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(s[0]), L.ptr(s[1]), L.ptr(dy), L.ptr(None), g, rpg, C, 1, st)),
            "bwd_app+res": (5, lambda: L.call("vc_bn_bwd_apply", 1, L.ptr(y), L.ptr(d), L.ptr(out), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(s[0]), L.ptr(s[1]), L.ptr(dy), L.ptr(dres), g, rpg, C, 1, st)),
        }
        row = f"{hw:>4}x{hw:<3}x{C:<4}{S:8.0f} |"
        for name, (passes, fn) in ks.items():
            ms = timeit(fn)
            row += f" {ms:6.3f}/{passes * S / ms:5.0f}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
