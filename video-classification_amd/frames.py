"""GPU-side input pipeline: decoded uint8 frames go to the device as they are and are resized, flipped and
normalised there by one kernel (``vc_frames_transform``), instead of per frame on the host with PIL + torchvision
as the reference does (utils/common/data_loaders.py:39-51, 88-100, 186-198; transform chain of train.py:344-388).

Host to device traffic drops from 602 KB (fp32, 3 x 224 x 224) to the raw frame (261 KB for 256 x 340 RGB) and the
data-loader workers only decode JPEG.  The uint8 result of the resize equals Pillow's bit for bit and the fp32
output equals ``ToTensor`` + ``Normalize`` bit for bit, so a model sees the same clips either way.

  FrameTransform     the transform of one (source size, target size) pair: coefficient tables on the device
  RawFrameDataset    the reference's Dataset_CRNN / Dataset_SwinCRNN / Dataset_3DCNN without the transform:
                     returns the decoded frames as uint8 [T, H, W, 3]
  DevicePrefetcher   wraps a loader of raw clips: pinned staging, copy and transform on a side stream, one batch
                     ahead of the training step
"""
import ctypes
import os

import numpy as np
import torch
from torch.utils import data

from . import _lib as L

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # train.py:383-384
IMAGENET_STD = (0.229, 0.224, 0.225)
VC_U8 = 2  # vc_frames_transform: planar uint8 output (the resized, flipped frame before ToTensor)


def resample_coeffs(in_size, out_size):
    """Window bounds int32 [out_size, 2] and fixed-point weights int32 [out_size, ksize] (host arrays) from the
    library's host routine."""
    lib = L.lib()
    ks = lib.vc_resample_ksize(int(in_size), int(out_size))
    if ks < 0:
        raise ValueError(f"resample {in_size} -> {out_size}: bad sizes")
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ks), np.int32)
    rc = lib.vc_resample_coeffs(int(in_size), int(out_size), bounds.ctypes.data_as(ctypes.c_void_p),
                                kk.ctypes.data_as(ctypes.c_void_p))
    if rc != ks:
        L.check(rc if rc < 0 else -4, "vc_resample_coeffs")
    return bounds, kk


def _jitter_range(v, center=1.0, clip_zero=True):
    """torchvision's ColorJitter argument check: a number v means [center - v, center + v] (brightness / contrast /
    saturation clipped at 0), a pair is taken as it is; None or a zero-width range around the neutral value = off."""
    if v is None:
        return None
    lo, hi = (center - float(v), center + float(v)) if not isinstance(v, (tuple, list)) else (float(v[0]), float(v[1]))
    if clip_zero:
        lo = max(lo, 0.0)
    return None if lo == hi == center else (lo, hi)


def draw_jitter(n, brightness=None, contrast=None, saturation=None, hue=None, generator=None):
    """Per-frame ColorJitter parameters as torchvision draws them per call (a permutation of the four operations,
    one uniform factor each): (order uint8 [n, 4], factors float32 [n, 3], hue_shift uint8 [n]) on the host.
    An operation that is switched off gets code 255 in `order`."""
    rb, rc, rs = _jitter_range(brightness), _jitter_range(contrast), _jitter_range(saturation)
    rh = _jitter_range(hue, center=0.0, clip_zero=False)
    order = torch.argsort(torch.rand(n, 4, generator=generator), dim=1).to(torch.uint8)
    factors = torch.ones(n, 3, dtype=torch.float32)
    for j, r in enumerate((rb, rc, rs)):
        if r is None:
            order[order == j] = 255
        else:
            factors[:, j] = r[0] + (r[1] - r[0]) * torch.rand(n, generator=generator)
    shift = torch.zeros(n, dtype=torch.uint8)
    if rh is None:
        order[order == 3] = 255
    else:
        h = (rh[0] + (rh[1] - rh[0]) * torch.rand(n, generator=generator)).double()
        shift = ((h * 255).trunc().to(torch.int64) & 255).to(torch.uint8)  # np.uint8(hue_factor * 255)
    return order, factors, shift


class FrameTransform:
    """Resize([size_h, size_w]) -> per-frame flip -> [ColorJitter] -> ToTensor -> Normalize(mean, std) of uint8 RGB
    frames on the device; ``gray=True`` converts to luma first (Dataset_3DCNN's ``convert('L')``) and yields one
    channel.  ``jitter=dict(brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1)`` (train.py:377-378) draws the
    ColorJitter parameters per frame, as the reference's per-image transform call does."""

    def __init__(self, size, mean=IMAGENET_MEAN, std=IMAGENET_STD, gray=False, device="cuda", dtype=torch.float32,
                 jitter=None, generator=None):
        if jitter and gray:
            raise ValueError("ColorJitter applies to colour frames (the reference switches it off for Conv3D)")
        self.jitter, self.generator = dict(jitter) if jitter else None, generator
        self.hd, self.wd = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        self.gray = bool(gray)
        self.device = torch.device(device)
        self.dtype = dtype
        c = 1 if gray else 3
        mean, std = list(mean), list(std)
        if len(mean) != c or len(std) != c:
            raise ValueError(f"mean / std need {c} entries")
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(std, dtype=torch.float32, device=self.device)
        self._tables = {}

    def _tab(self, hs, ws):
        key = (hs, ws)
        if key not in self._tables:
            xb, xk = resample_coeffs(ws, self.wd)
            yb, yk = resample_coeffs(hs, self.hd)
            dev = lambda a: torch.from_numpy(a).to(self.device)
            self._tables[key] = (dev(xb), dev(xk), xk.shape[1], dev(yb), dev(yk), yk.shape[1])
        return self._tables[key]

    def __call__(self, frames, flip=None, out=None, jitter_params=None):
        """frames uint8 [..., H, W, 3] on the device -> [..., C, size_h, size_w]; flip: uint8 / bool, one entry
        per frame (leading dimensions flattened), nonzero = mirror that frame; jitter_params: (order, factors,
        hue_shift) as `draw_jitter` returns them, drawn here if the transform was built with ``jitter=``."""
        if frames.device.type != "cuda":
            raise RuntimeError("FrameTransform runs on the GPU (there is no CPU fallback); move the frames first")
        if frames.dtype != torch.uint8 or frames.dim() < 3 or frames.shape[-1] != 3:
            raise ValueError(f"expected uint8 [..., H, W, 3] frames, got {frames.dtype} {tuple(frames.shape)}")
        frames = frames.contiguous()
        lead = frames.shape[:-3]
        hs, ws = int(frames.shape[-3]), int(frames.shape[-2])
        n = int(np.prod(lead)) if len(lead) else 1
        c = 1 if self.gray else 3
        if out is None:
            out = torch.empty(*lead, c, self.hd, self.wd, device=frames.device, dtype=self.dtype)
        elif out.numel() != n * c * self.hd * self.wd or not out.is_contiguous():
            raise ValueError("out has the wrong size or is not contiguous")
        if flip is not None:
            flip = flip.to(device=frames.device, dtype=torch.uint8).contiguous()
            if flip.numel() != n:
                raise ValueError(f"flip needs one entry per frame ({n}), got {flip.numel()}")
        xb, xk, ksx, yb, yk, ksy = self._tab(hs, ws)
        if jitter_params is None and self.jitter:
            jitter_params = draw_jitter(n, generator=self.generator, **self.jitter)
        if jitter_params is None:
            L.call("vc_frames_transform", L.ptr(frames), n, hs, ws, self.gray, self.hd, self.wd, L.ptr(xb), L.ptr(xk),
                   ksx, L.ptr(yb), L.ptr(yk), ksy, L.ptr(flip), L.ptr(self.mean), L.ptr(self.std), L.ptr(out),
                   L.dtype_code(out.dtype), L.stream())
            return out
        if self.gray:
            raise ValueError("ColorJitter applies to colour frames")
        order, factors, shift = (t.to(frames.device, non_blocking=True).contiguous() for t in jitter_params)
        if order.numel() != 4 * n or factors.numel() != 3 * n or shift.numel() != n or order.dtype != torch.uint8 \
                or factors.dtype != torch.float32 or shift.dtype != torch.uint8:
            raise ValueError("jitter_params: order uint8 [n, 4], factors float32 [n, 3], hue_shift uint8 [n]")
        planes = torch.empty(n, 3, self.hd, self.wd, device=frames.device, dtype=torch.uint8)
        L.call("vc_frames_transform", L.ptr(frames), n, hs, ws, 0, self.hd, self.wd, L.ptr(xb), L.ptr(xk), ksx,
               L.ptr(yb), L.ptr(yk), ksy, L.ptr(flip), L.ptr(None), L.ptr(None), L.ptr(planes), VC_U8, L.stream())
        L.call("vc_frames_jitter", L.ptr(planes), n, self.hd, self.wd, L.ptr(order), L.ptr(factors), L.ptr(shift),
               L.ptr(self.mean), L.ptr(self.std), L.ptr(out), L.dtype_code(out.dtype), L.stream())
        return out


class RawFrameDataset(data.Dataset):
    """Same constructor and file naming as the reference's datasets (data_loaders.py:62-108), without the
    transform: ``__getitem__`` returns the decoded frames uint8 [T, H, W, 3] and the label LongTensor [1].  The
    workers are left with JPEG decoding only; resize / flip / normalise run on the device."""

    def __init__(self, data_path, folders, labels, frames, transform=None):
        if transform is not None:
            raise ValueError("RawFrameDataset hands out raw frames; give the transform to DevicePrefetcher")
        self.data_path, self.folders, self.labels, self.frames = data_path, folders, labels, frames

    def __len__(self):
        return len(self.folders)

    def read_images(self, path, selected_folder):
        from PIL import Image
        X = []
        for i in self.frames:
            with Image.open(os.path.join(path, selected_folder, "frame{:06d}.jpg".format(i))) as im:
                X.append(torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.uint8).copy()))
        return torch.stack(X, dim=0)

    def __getitem__(self, index):
        return self.read_images(self.data_path, self.folders[index]), torch.LongTensor([self.labels[index]])


def raw_collate(batch):
    """Clips of one batch may come from videos of different frame sizes: keep a list, stack later per size."""
    return [b[0] for b in batch], torch.stack([b[1] for b in batch], dim=0)


class DevicePrefetcher:
    """Iterates a loader of raw clips (``RawFrameDataset`` + ``raw_collate``, or any loader that yields a uint8
    tensor [B, T, H, W, 3] / a list of [T, H, W, 3] tensors and the labels) and yields what the reference's loaders
    yield: (clip float [B, T, 3, S, S] -- [B, 1, T, S, S] with a gray transform --, target int64 [B, 1]), both on
    the device.  The copy and the transform of batch i + 1 run on a side stream while batch i trains.

    ``flip_p``: probability of mirroring a frame, drawn per FRAME as the reference's per-image transform call does
    (0 for validation); ``generator`` makes the draws reproducible."""

    def __init__(self, loader, transform, flip_p=0.0, generator=None):
        self.loader, self.tf, self.flip_p, self.gen = loader, transform, float(flip_p), generator
        self.stream = torch.cuda.Stream(device=transform.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, clips, target):
        dev = self.tf.device
        if torch.is_tensor(clips):
            clips = list(clips.unbind(0))
        B, T = len(clips), int(clips[0].shape[0])
        c = 1 if self.tf.gray else 3
        flip = None
        if self.flip_p > 0:
            flip = (torch.rand(B * T, generator=self.gen) < self.flip_p).to(torch.uint8)
        with torch.cuda.stream(self.stream):
            out = torch.empty(B, T, c, self.tf.hd, self.tf.wd, device=dev, dtype=self.tf.dtype)
            fl = flip.pin_memory().to(dev, non_blocking=True).view(B, T) if flip is not None else None
            sizes = {}
            for i, cl in enumerate(clips):
                sizes.setdefault(tuple(cl.shape), []).append(i)
            for shape, idx in sizes.items():
                host = torch.stack([clips[i] for i in idx], dim=0).pin_memory()
                raw = host.to(dev, non_blocking=True)
                if len(idx) == B:
                    self.tf(raw, fl, out=out)
                else:
                    sel = torch.tensor(idx, device=dev)
                    part = self.tf(raw, fl[sel] if fl is not None else None)
                    out.index_copy_(0, sel, part)
                raw.record_stream(self.stream)
            if self.tf.gray:  # Dataset_3DCNN: [1, T, H, W] per clip (data_loaders.py:48-56)
                out = out.transpose(1, 2).contiguous()
            tgt = target.pin_memory().to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, tgt, ev

    def __iter__(self):
        nxt = None
        for clips, target in self.loader:
            cur, nxt = nxt, self._stage(clips, target)
            if cur is not None:
                yield self._hand_over(cur)
        if nxt is not None:
            yield self._hand_over(nxt)

    def _hand_over(self, staged):
        out, tgt, ev = staged
        cur = torch.cuda.current_stream(self.tf.device)
        cur.wait_event(ev)
        out.record_stream(cur)
        tgt.record_stream(cur)
        return out, tgt
