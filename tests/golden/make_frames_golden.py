#!/usr/bin/env python3
"""Generate tests/golden/frames_*.npz: the reference's per-frame transform chain (train.py:344-388, applied per
frame by utils/common/data_loaders.py:39-51, 88-100) run with the libraries the reference itself calls.

torchvision is not installed in this image, so its three transforms are spelled out with what they dispatch to
for a PIL input: Resize -> ``Image.resize((w, h), BILINEAR)``, RandomHorizontalFlip (when it fires) ->
``Image.transpose(FLIP_LEFT_RIGHT)``, ToTensor -> uint8 HWC to float32 CHW divided by 255, Normalize ->
``sub_(mean).div_(std)`` in float32.  Pillow (12.2.0 here) does the integer work, torch the float tail.

    python tests/golden/make_frames_golden.py
"""
import os

import numpy as np
import torch
from PIL import Image

OUT = os.path.dirname(os.path.abspath(__file__))


def chain(frame, size, mean, std, flip, gray):
    im = Image.fromarray(frame)
    if gray:
        im = im.convert("L")
    im = im.resize((size[1], size[0]), Image.BILINEAR)
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    a = np.asarray(im)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    return np.asarray(im).copy(), t.sub_(m).div_(s).numpy()


def make(name, n, hs, ws, size, gray, seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    # half of the frames smooth (what video looks like), half noise (every rounding case)
    yy, xx = np.mgrid[0:hs, 0:ws]
    for i in range(0, n, 2):
        for c in range(3):
            frames[i, :, :, c] = ((np.sin(xx / (3.0 + c) + i) + np.cos(yy / (5.0 - c))) * 63 + 128).clip(0, 255)
    flip = (rng.random(n) < 0.5).astype(np.uint8)
    flip[0], flip[-1] = 0, 1
    mean, std = ([0.5], [0.5]) if gray else ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    u8, f32 = zip(*[chain(frames[i], size, mean, std, flip[i], gray) for i in range(n)])
    np.savez_compressed(os.path.join(OUT, name), frames=frames, flip=flip, size=np.asarray(size),
                        mean=np.asarray(mean, np.float32), std=np.asarray(std, np.float32),
                        resized_u8=np.stack(u8), out=np.stack(f32), gray=np.asarray(int(gray)))
    print(name, frames.shape, "->", np.stack(f32).shape)


def tv_color_jitter(im, order, b, c, s, h):
    """torchvision's ColorJitter.forward on a PIL image with the drawn parameters: ImageEnhance for brightness /
    contrast / saturation, the HSV round trip of adjust_hue (uint8 H plus np.uint8(hue_factor * 255), wrapping)."""
    from PIL import ImageEnhance
    for op in order:
        if op == 0:
            im = ImageEnhance.Brightness(im).enhance(b)
        elif op == 1:
            im = ImageEnhance.Contrast(im).enhance(c)
        elif op == 2:
            im = ImageEnhance.Color(im).enhance(s)
        elif op == 3:
            hh, ss, vv = im.convert("HSV").split()
            np_h = np.array(hh, dtype=np.uint8)
            with np.errstate(over="ignore"):
                np_h += np.array(h * 255).astype(np.uint8)
            im = Image.merge("HSV", (Image.fromarray(np_h, "L"), ss, vv)).convert("RGB")
    return im


def make_jitter(name, n, hs, ws, size, seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:hs, 0:ws]
    for i in range(0, n, 2):
        for c in range(3):
            frames[i, :, :, c] = ((np.sin(xx / (3.0 + c) + i) + np.cos(yy / (5.0 - c))) * 63 + 128).clip(0, 255)
    flip = (rng.random(n) < 0.5).astype(np.uint8)
    order = np.stack([rng.permutation(4) for _ in range(n)]).astype(np.uint8)
    factors = rng.uniform(0.8, 1.2, (n, 3)).astype(np.float32)      # train.py:378: 0.2 / 0.2 / 0.2
    factors[-1] = (0.0, 2.3, 1.7)                                    # outside [0, 1]: the clipped branch of blend
    hue = rng.uniform(-0.1, 0.1, n).astype(np.float32)               # hue 0.1
    hue[-2] = -0.5
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    u8, f32 = [], []
    for i in range(n):
        im = Image.fromarray(frames[i]).resize((size[1], size[0]), Image.BILINEAR)
        if flip[i]:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        im = tv_color_jitter(im, order[i], float(factors[i, 0]), float(factors[i, 1]), float(factors[i, 2]),
                             float(hue[i]))
        a = np.asarray(im).copy()
        t = torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
        s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
        u8.append(a)
        f32.append(t.sub_(m).div_(s).numpy())
    np.savez_compressed(os.path.join(OUT, name), frames=frames, flip=flip, size=np.asarray(size), order=order,
                        factors=factors, hue=hue, mean=np.asarray(mean, np.float32), std=np.asarray(std, np.float32),
                        jittered_u8=np.stack(u8), out=np.stack(f32))
    print(name, frames.shape, "->", np.stack(f32).shape)


if __name__ == "__main__":
    make_jitter("jitter_color.npz", 6, 48, 64, (40, 40), 5)          # Resize -> flip -> ColorJitter -> ToTensor -> Normalize
    make("frames_color_down.npz", 4, 64, 85, (56, 56), False, 1)     # jpegs_256-like aspect, 256x340 -> 224x224
    make("frames_gray_down.npz", 4, 60, 80, (28, 28), True, 2)       # Conv3D: luma, 240x320 -> 112x112
    make("frames_color_mixed.npz", 2, 23, 90, (40, 40), False, 3)    # up-scaling rows, 2.25x reduction of columns
    make("frames_color_same.npz", 2, 32, 32, (32, 32), False, 4)     # already the target size
