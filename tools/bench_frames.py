#!/usr/bin/env python3
"""Throughput of the GPU-side frame transform (vc_frames_transform) against the HBM roofline, with the host chain
the reference runs per frame (Pillow resize + the float tail, utils/common/data_loaders.py:88-100 with
train.py:367-388) timed beside it on one core, which is what one DataLoader worker gives.

    python tools/bench_frames.py [--frames 512] [--src 256x340] [--size 224] [--gray] [--dtype f32|bf16]

Prints one JSON line.  Algorithmic bytes per frame = the uint8 source frame + the output frame."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from video_classification_amd import frames as FR  # noqa: E402


def host_chain(frames, size, mean, std, gray):
    """What a DataLoader worker of the reference does with decoded frames (JPEG decoding not included)."""
    from PIL import Image
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    outs = []
    for f in frames:
        im = Image.fromarray(f)
        if gray:
            im = im.convert("L")
        im = im.resize((size, size), Image.BILINEAR)
        a = np.asarray(im)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        outs.append(t.sub_(m).div_(s))
    return torch.stack(outs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--src", default="256x340")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--gray", action="store_true")
    ap.add_argument("--jitter", action="store_true", help="ColorJitter(0.2, 0.2, 0.2, 0.1) between flip and ToTensor")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-frames", type=int, default=256)
    a = ap.parse_args()
    hs, ws = (int(v) for v in a.src.split("x"))
    dev = torch.device("cuda:0")
    dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
    mean, std = ([0.5], [0.5]) if a.gray else (FR.IMAGENET_MEAN, FR.IMAGENET_STD)
    rng = np.random.default_rng(0)
    host = rng.integers(0, 256, (a.frames, hs, ws, 3), dtype=np.uint8)
    frames = torch.from_numpy(host).to(dev)
    flip = torch.from_numpy((rng.random(a.frames) < 0.5).astype(np.uint8)).to(dev)
    tf = FR.FrameTransform(a.size, mean, std, gray=a.gray, device=dev, dtype=dt)
    jp = None
    if a.jitter:  # parameters drawn once and resident, so that the timed region holds the two kernels only
        jp = tuple(t.to(dev) for t in FR.draw_jitter(a.frames, 0.2, 0.2, 0.2, 0.1, generator=torch.Generator().manual_seed(0)))
    out = tf(frames, flip, jitter_params=jp)
    for _ in range(5):
        tf(frames, flip, out=out, jitter_params=jp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()  # the kernel is launched on torch's current stream, which is the one these events time
    for _ in range(a.iters):
        tf(frames, flip, out=out, jitter_params=jp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    c = 1 if a.gray else 3
    bytes_per_frame = hs * ws * 3 + c * a.size * a.size * out.element_size()
    achieved = a.frames * bytes_per_frame / (ms * 1e-3) / 1e9

    # host to device copy of the raw frames from pinned memory, for the PCIe-inclusive rate
    pinned = torch.from_numpy(host).pin_memory()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        frames.copy_(pinned, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    h2d_ms = e0.elapsed_time(e1) / 5

    line = {"metric": "frames_transformed_per_sec", "value": a.frames / (ms * 1e-3), "unit": "frames/s", "n_gpus": 1,
            "ms_per_call": ms, "dtype": "u8 -> " + a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.frames} RGB frames {hs}x{ws} -> {c}x{a.size}x{a.size}"
                                   f"{' (luma)' if a.gray else ''}, resize + flip{' + ColorJitter' if a.jitter else ''} + normalise"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "algorithmic_bytes_per_frame": bytes_per_frame},
            "h2d_ms_raw_frames": h2d_ms, "with_h2d_frames_per_sec": a.frames / ((ms + h2d_ms) * 1e-3)}
    if a.jitter:
        line["cpu_baseline"] = None  # the host chain of this tool has no ColorJitter (torchvision is absent)
        print(json.dumps(line))
        return
    try:
        torch.set_num_threads(1)
        n = min(a.cpu_frames, a.frames)
        host_chain(host[:8], a.size, mean, std, a.gray)
        t0 = time.perf_counter()
        ref = host_chain(host[:n], a.size, mean, std, a.gray)
        dtc = time.perf_counter() - t0
        same = bool(torch.equal(ref, tf(frames[:n]).float().cpu())) if dt == torch.float32 else None
        line["cpu_baseline"] = {"value": n / dtc, "unit": "frames/s", "cores": 1, "kind": "reference",
                                "sample": f"{n} frames through Pillow {__import__('PIL').__version__} resize + "
                                          "ToTensor + Normalize on one core (one DataLoader worker)",
                                "bit_identical_to_gpu": same}
    except ImportError:
        line["cpu_baseline"] = None
    print(json.dumps(line))


if __name__ == "__main__":
    main()
