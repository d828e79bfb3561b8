"""Time of the width-unfold pass of the ResNet stem (512 frames of 3 x 224 x 224 fp32 -> [512][224][112][24] bf16) and of
the stem's max-pool backward.  VC_UNFOLD_ROWS (image rows per block) is read once per process."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from video_classification_amd import _lib as L  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B, T, C, H, W, kw, sw, pw, Jp = 32, 16, 3, 224, 224, 7, 2, 3, 24
    x = torch.randn(B, T, C, H, W, device="cuda")
    Wo = (W + 2 * pw - kw) // sw + 1
    U = torch.empty(T * B, 1, H, Wo, Jp, device="cuda", dtype=torch.bfloat16)
    ms = timed(lambda: L.call("vc_clip_unfold_w", 1, L.ptr(x), L.ptr(U), B, T, C, H, W, kw, sw, pw, Jp, 1, 0, L.stream()))
    gb = (x.numel() * 4 + U.numel() * 2) / 1e9
    print(f"unfold: {ms * 1e3:.1f} us, {gb / ms:.2f} TB/s  (VC_UNFOLD_ROWS={os.environ.get('VC_UNFOLD_ROWS', 'default')})")
    N, Hh, Ww, Cc = 512, 112, 112, 64
    a = torch.randn(N, Hh, Ww, Cc, device="cuda").bfloat16()
    y = torch.empty(N, 56, 56, Cc, device="cuda", dtype=torch.bfloat16)
    idx = torch.empty(N, 56, 56, Cc, device="cuda", dtype=torch.uint8)
    ms = timed(lambda: L.call("vc_maxpool2d_fwd", 1, L.ptr(a), L.ptr(y), L.ptr(idx), N, Hh, Ww, Cc, 3, 2, 1, L.stream()))
    gb = (a.numel() * 2 + y.numel() * 3) / 1e9
    print(f"maxpool fwd: {ms * 1e3:.1f} us, {gb / ms:.2f} TB/s")
    dy = torch.randn_like(y)
    dx = torch.empty_like(a)
    ms = timed(lambda: L.call("vc_maxpool2d_bwd", 1, L.ptr(dy), L.ptr(idx), L.ptr(dx), N, Hh, Ww, Cc, 3, 2, 1, L.stream()))
    print(f"maxpool bwd: {ms * 1e3:.1f} us, {gb / ms:.2f} TB/s")


if __name__ == "__main__":
    main()
