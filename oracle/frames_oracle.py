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

  * ColorJitter(brightness, contrast, saturation, hue) (train.py:377-378), which torchvision applies to a PIL image
    as ImageEnhance.Brightness / Contrast / Color and an HSV round trip, in a random order drawn per call:
    ``Image.blend`` in float32 (in1 + alpha * (in2 - in1), truncated; clipped when alpha leaves [0, 1]) against
    black / the rounded mean luma / the pixel's luma, and ``Convert.c``'s rgb2hsv / hsv2rgb (float and double
    steps as in the C source, uint8 H shifted with wrap-around).

Pinning: ``tests/test_frames.py`` holds this file bit-exact to Pillow itself wherever Pillow imports (this image,
here and on the GPU box) -- the resize on six geometries, blend on all 65536 byte pairs, both HSV conversions on all
2^24 triples -- and to the fixtures ``tests/golden/frames_*.npz`` that ``tests/golden/make_frames_golden.py``
produced with Pillow + torch.
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


# ---- ColorJitter ------------------------------------------------------------------------------------------------
def blend_u8(in1, in2, alpha):
    """Pillow's ``Image.blend(im1, im2, alpha)`` on uint8 arrays: float32 arithmetic, truncation."""
    al = np.float32(alpha)
    a = in1.astype(np.int32)
    s = a.astype(np.float32) + al * (in2.astype(np.int32) - a).astype(np.float32)
    if 0.0 <= float(al) <= 1.0:
        return s.astype(np.int32).astype(np.uint8)
    return np.where(s <= 0, 0, np.where(s >= 255, 255, s.astype(np.int32))).astype(np.uint8)


def rgb_to_hsv_u8(x):
    """Pillow's ``convert('HSV')`` of uint8 RGB [..., 3]."""
    r, g, b = (x[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    with np.errstate(all="ignore"):
        s = cr / maxc.astype(np.float32)
        rc = (maxc - r).astype(np.float32) / cr
        gc = (maxc - g).astype(np.float32) / cr
        bc = (maxc - b).astype(np.float32) / cr
        d = np.float64
        h = np.where(r == maxc, (bc - gc).astype(d),
                     np.where(g == maxc, 2.0 + rc.astype(d) - bc.astype(d), 4.0 + gc.astype(d) - rc.astype(d)))
        h = np.fmod(h.astype(np.float32).astype(d) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((h.astype(d) * 255.0).astype(np.int32), 0, 255)
        us = np.clip((s.astype(d) * 255.0).astype(np.int32), 0, 255)
    gray = maxc == minc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb_u8(x):
    """Pillow's ``convert('RGB')`` of a uint8 HSV image [..., 3]."""
    d = np.float64
    hh = x[..., 0].astype(np.float32).astype(d) * 6.0 / 255.0
    i = np.floor(hh).astype(np.int32)
    f = (hh - i.astype(np.float32).astype(d)).astype(np.float32).astype(d)
    fs = (x[..., 1].astype(np.float32).astype(d) / 255.0).astype(np.float32).astype(d)
    v = x[..., 2].astype(np.int32)
    vf = v.astype(d)

    def rnd(z):  # C round(): half away from zero
        return np.clip(np.where(z >= 0, np.floor(z + 0.5), np.ceil(z - 0.5)).astype(np.int32), 0, 255)

    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    k = i % 6
    R = np.choose(k, [v, q, p, p, t, v])
    G = np.choose(k, [t, v, v, q, p, p])
    B = np.choose(k, [p, p, t, v, v, q])
    g0 = x[..., 1] == 0
    return np.stack([np.where(g0, v, R), np.where(g0, v, G), np.where(g0, v, B)], -1).astype(np.uint8)


def hue_shift_byte(hue_factor):
    """What torchvision adds to the uint8 H channel: ``np.uint8(hue_factor * 255)`` (truncation, wrap-around)."""
    return int(hue_factor * 255) & 255


def color_jitter_u8(img, order, brightness, contrast, saturation, hue_factor):
    """torchvision's ColorJitter on a PIL RGB image [H, W, 3] with the drawn parameters: `order` is the drawn
    permutation of (0 brightness, 1 contrast, 2 saturation, 3 hue); a factor of None skips that operation."""
    a = img
    for op in order:
        if op == 0 and brightness is not None:
            a = blend_u8(np.zeros_like(a), a, brightness)
        elif op == 1 and contrast is not None:
            lum = rgb_to_l(a)
            hist = np.bincount(lum.reshape(-1), minlength=256)
            mean = int(float(int((hist * np.arange(256)).sum())) / float(lum.size) + 0.5)
            a = blend_u8(np.full_like(a, mean), a, contrast)
        elif op == 2 and saturation is not None:
            a = blend_u8(np.repeat(rgb_to_l(a)[..., None], 3, -1), a, saturation)
        elif op == 3 and hue_factor is not None:
            hsv = rgb_to_hsv_u8(a)
            hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_byte(hue_factor)) & 255
            a = hsv_to_rgb_u8(hsv)
    return a


def transform_frames_jitter(frames, out_h, out_w, mean, std, flip, params):
    """transform_frames with ColorJitter between the flip and ToTensor (the order of train.py:367-388); params[n] =
    (order, brightness, contrast, saturation, hue_factor) as drawn for frame n."""
    mean = np.asarray(mean, np.float32)
    std = np.asarray(std, np.float32)
    outs = []
    for n in range(frames.shape[0]):
        r = resize_bilinear_u8(frames[n], out_h, out_w)
        if flip is not None and flip[n]:
            r = r[:, ::-1]
        r = color_jitter_u8(np.ascontiguousarray(r), *params[n])
        t = r.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        outs.append((t - mean[:, None, None]) / std[:, None, None])
    return np.stack(outs).astype(np.float32)
