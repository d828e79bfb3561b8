#!/usr/bin/env python3
"""Per-layer micro-benchmark of the implicit-GEMM kernels on the convolution shapes of the bench workload
(ResNet-50 trunk at 512 frames of 224x224, plus the heads / LSTM GEMMs).  Prints algorithmic TFLOP/s
(unpadded channels) for forward, data gradient and weight gradient of every distinct shape.

    python tools/bench_conv.py [--frames 512] [--reps 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from video_classification_amd import _lib as L  # noqa: E402


def resnet50_shapes(frames):
    """(name, N, H, W, Ci, Co, k, s, p, count)"""
    shapes = [("stem7x7", frames, 224, 224, 3, 64, 7, 2, 3, 1)]
    inpl, H = 64, 56
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for b in range(n):
            s = (1 if li == 0 else 2) if b == 0 else 1
            shapes.append((f"l{li + 1}b{b}.c1", frames, H, H, inpl, planes, 1, 1, 0, 1))
            shapes.append((f"l{li + 1}b{b}.c2", frames, H, H, planes, planes, 3, s, 1, 1))
            Ho = H // s
            shapes.append((f"l{li + 1}b{b}.c3", frames, Ho, Ho, planes, planes * 4, 1, 1, 0, 1))
            if b == 0:
                shapes.append((f"l{li + 1}b{b}.ds", frames, H, H, inpl, planes * 4, 1, s, 0, 1))
            inpl, H = planes * 4, Ho
    # merge identical shapes
    merged = {}
    for sh in shapes:
        key = sh[1:9]
        if key in merged:
            merged[key][-1] += 1
        else:
            merged[key] = list(sh)
    return [tuple(v) for v in merged.values()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default=None, help="run only the layers with these names, comma separated (for counter collection)")
    ap.add_argument("--passes", default="fwd,dgrad,wgrad")
    ap.add_argument("--stats", action="store_true",
                    help="forward pass with the BatchNorm slab sums in its epilogue (vc_conv_fwd_stats), as in a training step")
    ap.add_argument("--flush", action="store_true",
                    help="overwrite a 1 GiB buffer before every timed launch: operands come from HBM as inside a "
                         "training step, not from the 256 MB infinity cache the previous repetition left warm")
    args = ap.parse_args()
    passes = args.passes.split(",")
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    flush_buf = torch.zeros(256 << 20, device=dev) if args.flush else None
    print(f"{'layer':12s} {'N,H,W,Ci->Co k s':34s} {'cnt':>3s} {'GF':>8s} | {'fwd TF':>7s} {'ms':>6s} | {'dgrad TF':>8s} {'ms':>6s} |"
          f" {'wgrad TF':>8s} {'ms':>6s}")
    for name, N, H, W, Ci, Co, k, s, p, cnt in resnet50_shapes(args.frames):
        if args.only and name not in args.only.split(","):
            continue
        cip = (Ci + 7) // 8 * 8
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        desc = L.int_array([N, 1, H, W, cip, 1, Ho, Wo, Co, 1, k, k, 1, s, s, 0, p, p])
        x = torch.randn(N, H, W, cip, device=dev).to(dt)
        dy = torch.randn(N, Ho, Wo, Co, device=dev).to(dt)
        taps = k * k
        wf = (torch.randn(Co, taps, cip, device=dev) * 0.05).to(dt)
        wt = (torch.randn(Ci, taps, Co, device=dev) * 0.05).to(dt)
        y = torch.empty(N, Ho, Wo, Co, device=dev, dtype=dt)
        dx = torch.empty(N, H, W, cip, device=dev, dtype=dt)
        dwp = torch.zeros(Co, taps, cip, device=dev, dtype=torch.float32)
        flops = 2.0 * N * Ho * Wo * Co * Ci * taps
        st = L.stream()

        def run(fn):
            fn()
            torch.cuda.synchronize()
            if args.flush:
                total = 0.0
                for _ in range(args.reps):
                    flush_buf.add_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    total += e0.elapsed_time(e1)
                return total / args.reps
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.reps

        t_f = t_d = t_w = float("nan")
        if "fwd" in passes:
            if args.stats and L.lib().vc_conv_fwd_stats_supported(desc, 1, 0, 0):
                part = torch.empty(N * Ho * Wo // 64, 2, Co, device=dev, dtype=torch.float32)
                t_f = run(lambda: L.call("vc_conv_fwd_stats", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y), 0, 0,
                                         L.ptr(part), st))
            else:
                t_f = run(lambda: L.call("vc_conv_fwd", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y), 0, 0, 0, 0, st))
        if Ci % 8 == 0 and "dgrad" in passes:
            t_d = run(lambda: L.call("vc_conv_dgrad", desc, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx), 0, 0, 0, st))
        if "wgrad" in passes:
            t_w = run(lambda: L.call("vc_conv_wgrad", desc, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dwp), L.ptr(None), 0, st))
        tf = lambda t: flops / (t * 1e-3) / 1e12  # noqa: E731
        print(f"{name:12s} {f'{N},{H},{W},{Ci}->{Co} k{k} s{s}':34s} {cnt:3d} {flops / 1e9:8.1f} | {tf(t_f):7.0f} {t_f:6.3f} | "
              f"{tf(t_d):8.0f} {t_d:6.3f} | {tf(t_w):8.0f} {t_w:6.3f}")
        for key, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            if t == t:
                tot[key][0] += flops * cnt
                tot[key][1] += t * cnt
    for key, (fl, ms) in tot.items():
        if ms > 0:
            print(f"total {key:6s}: {ms:7.2f} ms per step, {fl / (ms * 1e-3) / 1e12:6.0f} TFLOP/s")


if __name__ == "__main__":
    main()
