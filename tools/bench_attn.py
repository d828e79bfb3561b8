#!/usr/bin/env python3
"""Window attention (Swin-T stage shapes) forward and backward through the C ABI: time per launch, or -- under
rocprofv3 --pmc (tools/pmc_attn.sh) -- a couple of launches for the counters.

    python tools/bench_attn.py [--frames 128] [--stage 1] [--shift 3] [--reps 10] [--passes fwd,bwd]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from video_classification_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128, help="8 clips x 16 frames")
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2, 3, 4])
    ap.add_argument("--shift", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--passes", default="fwd,bwd")
    a = ap.parse_args()
    side = 56 >> (a.stage - 1)
    C = 96 << (a.stage - 1)
    heads, ws = C // 32, 7
    B = a.frames
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * side * side, 3 * C, generator=g).to(dev, torch.bfloat16)
    dout = torch.randn(B * side * side, C, generator=g).to(dev, torch.bfloat16)
    table = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.02).to(dev)
    out = torch.empty_like(dout)
    dqkv = torch.empty_like(qkv)
    dtab = torch.zeros_like(table)
    scale = 32 ** -0.5
    shift = a.shift if side > ws else 0

    def fwd():
        L.call("vc_window_attn_fwd", 1, L.ptr(qkv), L.ptr(table), L.ptr(out), B, side, side, C, heads, ws, shift,
               float(scale), L.stream())

    def bwd():
        L.call("vc_window_attn_bwd", 1, L.ptr(qkv), L.ptr(dout), L.ptr(table), L.ptr(dqkv), L.ptr(dtab), B, side, side, C,
               heads, ws, shift, float(scale), L.stream())
    units = B * (side // ws) ** 2 * heads
    for name, fn, mfma in (("fwd", fwd, 24), ("bwd", bwd, 112)):
        if name not in a.passes.split(","):
            continue
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / a.reps * 1e3
        flops = units * mfma * 2 * 16 * 16 * 32
        print(f"window attention {name}: stage {a.stage} ({side}x{side}, {C} ch, {heads} heads), {B} frames, shift {shift}: "
              f"{us:8.1f} us per launch, {units} (window, head) units, {us * 1e3 / units * 4 * 256 / 1e3:6.2f} us per unit "
              f"and wave, {flops / us / 1e6:6.0f} TFLOP/s of MFMA work ({mfma} MFMAs per unit)")


if __name__ == "__main__":
    main()
