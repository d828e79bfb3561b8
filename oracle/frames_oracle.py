"""CPU oracle of the per-frame input transform: a numpy restatement, integer-exact.

TEST INFRASTRUCTURE ONLY.  Nothing under ``video_classification_amd`` imports this file; it is used by ``tests/``
and by the CPU-baseline leg of ``tools/bench_frames.py`` as the checker, never as the product path.

What the reference does to every decoded frame (paths relative to the reference checkout):

  Dataset_CRNN / Dataset_SwinCRNN.read_images   utils/common/data_loaders.py:88-100, 186-198
      Image.open(jpg) -> transform(image), one call PER FRAME (so a random flip is drawn per frame)
  Dataset_3DCNN.read_images                     utils/common/data_loaders.py:39-51
      Image.open(jpg).convert('L') -> transform(image)
  the transform                                 train.py:344-388
      Resize([S, S]) -> [RandomHorizontalFlip] -> [RandomCrop(S): a no-op after Resize to S x S]
      -> ToTensor -> Normalize(mean, std)       (colour: ImageNet mean / std; Conv3D: 0.5 / 0.5)

The arithmetic lives in third-party code that is absent from the reference checkout: torchvision
(``requirements.txt:5``: >=0.10.0, not installed here) hands ``Resize`` of a PIL image to Pillow's
``Image.resize(size, BILINEAR)``; ``ToTensor`` is uint8 HWC -> float32 CHW / 255; ``Normalize`` is
``(x - mean) / std`` in float32.  Pillow (``requirements.txt``: Pillow>=8.0; 12.2.0 in this image) is restated
below from its published algorithm (``src/libImaging/Resample.c``: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc; ``Convert.c``: rgb2l):

  * separable triangle filter whose support grows with the down-scaling factor (anti-aliased), window bounds and
    double-precision weights normalised per output pixel,
  * weights rounded to 22-bit fixed point, accumulation in int32 starting from 1 << 21, result >> 22 clipped to
    0..255, horizontal pass first, its uint8 result feeding the vertical pass,
  * RGB -> L:  (19595 R + 38470 G + 7471 B + 0x8000) >> 16.

Pinning: ``tests/test_frames_oracle.py`` holds this file bit-exact to Pillow itself wherever Pillow imports (this
image, here and on the GPU box) and to the fixtures ``tests/golden/frames_*.npz`` that
``tests/golden/make_frames_golden.py`` produced with Pillow + torch (ColorJitter is not restated: see DESIGN.md).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size):
    """Window bounds [out_size, 2] (first source index, tap count) and 22-bit fixed-point weights
    [out_size, ksize] of Pillow's bilinear (triangle) filter for a full-extent resize."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            for x in range(xmax):
                w[x] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass_rows(img, bounds, kk):
    """One resampling pass along axis 1 of a uint8 array [rows, n, channels]."""
    out = np.empty((img.shape[0], bounds.shape[0], img.shape[2]), np.uint8)
    src = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full((img.shape[0], img.shape[2]), 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(n):
            acc += src[:, x0 + k, :] * int(kk[xx, k])
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img, out_h, out_w):
    """Pillow's ``Image.resize((out_w, out_h), BILINEAR)`` of a uint8 image [H, W] or [H, W, C]."""
    a = img[:, :, None] if img.ndim == 2 else img
    h, w = a.shape[:2]
    if (h, w) != (out_h, out_w):
        if w != out_w:
            a = _pass_rows(a, *resample_coeffs(w, out_w))
        if h != out_h:
            a = _pass_rows(a.transpose(1, 0, 2), *resample_coeffs(h, out_h)).transpose(1, 0, 2)
    a = np.ascontiguousarray(a)
    return a[:, :, 0] if img.ndim == 2 else a


def rgb_to_l(img):
    """Pillow's ``convert('L')`` of an RGB uint8 image [H, W, 3] (ITU-R 601-2 luma, 16-bit fixed point)."""
    a = img.astype(np.int64)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def transform_frames(frames, out_h, out_w, mean, std, flip=None, gray=False):
    """frames uint8 [N, H, W, 3] -> float32 [N, C, out_h, out_w] (C = 1 if gray else 3): the reference's
    transform chain applied to every frame; flip[n] != 0 mirrors frame n after the resize."""
    mean = np.asarray(mean, np.float32)
    std = np.asarray(std, np.float32)
    outs = []
    for n in range(frames.shape[0]):
        f = frames[n]
        if gray:
            f = rgb_to_l(f)[:, :, None]
        r = resize_bilinear_u8(f, out_h, out_w)
        if flip is not None and flip[n]:
            r = r[:, ::-1]
        t = r.transpose(2, 0, 1).astype(np.float32) / np.float32(255)  # ToTensor
        outs.append((t - mean[:, None, None]) / std[:, None, None])    # Normalize
    return np.stack(outs).astype(np.float32)
