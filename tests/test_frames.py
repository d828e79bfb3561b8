"""Per-frame input transform (reference: train.py:344-388 applied per frame by utils/common/data_loaders.py:39-51,
88-100).  CPU tests pin the numpy oracle to the Pillow-made fixtures and to Pillow itself, and the library's host
coefficient routine to the oracle; GPU tests hold the HIP kernel to the oracle bit for bit."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from oracle import frames_oracle as FO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "frames_*.npz")))


def _load(path):
    z = np.load(path)  # plain arrays, no pickles
    return {k: z[k] for k in z.files}


def test_fixtures_present():
    assert len(FIXTURES) == 4 and os.path.exists(os.path.join(GOLD, "jitter_color.npz"))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_matches_pillow_fixtures(path):
    z = _load(path)
    gray = bool(z["gray"])
    h, w = (int(v) for v in z["size"])
    for i in range(z["frames"].shape[0]):
        f = FO.rgb_to_l(z["frames"][i]) if gray else z["frames"][i]
        r = FO.resize_bilinear_u8(f, h, w)
        if z["flip"][i]:
            r = r[:, ::-1]
        assert np.array_equal(r, z["resized_u8"][i])
    out = FO.transform_frames(z["frames"], h, w, z["mean"], z["std"], z["flip"], gray)
    assert out.dtype == np.float32 and np.array_equal(out, z["out"])  # bit-exact float32


@pytest.mark.parametrize("hs,ws,hd,wd", [(256, 340, 224, 224), (240, 320, 112, 112), (50, 70, 224, 224),
                                         (1080, 1920, 224, 224), (225, 223, 224, 224), (17, 9, 5, 3)])
def test_oracle_matches_pillow_live(hs, ws, hd, wd):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(hs * 7 + ws)
    a = rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
    assert np.array_equal(FO.resize_bilinear_u8(a, hd, wd), np.asarray(Image.fromarray(a).resize((wd, hd), Image.BILINEAR)))
    lum = np.asarray(Image.fromarray(a).convert("L"))
    assert np.array_equal(FO.rgb_to_l(a), lum)
    assert np.array_equal(FO.resize_bilinear_u8(lum, hd, wd),
                          np.asarray(Image.fromarray(lum).resize((wd, hd), Image.BILINEAR)))


@pytest.mark.parametrize("n_in,n_out", [(340, 224), (256, 224), (320, 112), (240, 112), (224, 224), (70, 224),
                                        (1920, 224), (9, 3), (1, 1), (1, 5), (1000, 1)])
def test_host_coefficients_match_oracle(n_in, n_out):
    from video_classification_amd import frames as FR
    b, k = FR.resample_coeffs(n_in, n_out)
    ob, ok = FO.resample_coeffs(n_in, n_out)
    assert k.shape == ok.shape
    assert np.array_equal(b, ob) and np.array_equal(k, ok)
    assert (k.sum(axis=1) - (1 << 22)).__abs__().max() <= k.shape[1]  # weights sum to one in fixed point


def test_transform_argument_checks():
    from video_classification_amd import _lib as L
    lib = L.lib()
    lib.vc_last_error.restype = ctypes.c_char_p
    rc = lib.vc_frames_transform(None, 2, 64, 85, 0, 56, 56, None, None, 3, None, None, 5, None, None, None, None, 0,
                                 None)
    assert rc == -4 and b"another geometry" in lib.vc_last_error()
    assert lib.vc_frames_transform(None, 0, 64, 85, 0, 56, 56, None, None, 5, None, None, 5, None, None, None, None,
                                   0, None) == 0  # no frames: nothing to do
    assert lib.vc_frames_transform(None, 2, 64, 85, 0, 56, 56, None, None, 5, None, None, 5, None, None, None, None,
                                   7, None) == -4
    from video_classification_amd import frames as FR
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FR.FrameTransform(8, device="cpu")(torch.zeros(1, 4, 4, 3, dtype=torch.uint8))


def test_raw_dataset_reads_what_pillow_decodes(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    from video_classification_amd import frames as FR
    rng = np.random.default_rng(0)
    for folder in ("v_A_g01_c01", "v_B_g02_c03"):
        os.makedirs(tmp_path / folder)
        for i in range(1, 6):
            Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).save(
                str(tmp_path / folder / f"frame{i:06d}.jpg"), quality=90)
    ds = FR.RawFrameDataset(str(tmp_path), ["v_A_g01_c01", "v_B_g02_c03"], [3, 7], [1, 3, 5])
    assert len(ds) == 2
    x, y = ds[1]
    assert x.dtype == torch.uint8 and tuple(x.shape) == (3, 24, 32, 3)
    assert y.dtype == torch.int64 and y.tolist() == [7]
    want = np.asarray(Image.open(str(tmp_path / "v_B_g02_c03" / "frame000003.jpg")))
    assert np.array_equal(x[1].numpy(), want)
    clips, target = FR.raw_collate([ds[0], ds[1]])
    assert len(clips) == 2 and tuple(target.shape) == (2, 1)


# ---- ColorJitter ------------------------------------------------------------------------------------------------
def _jitter_fixture():
    z = _load(os.path.join(GOLD, "jitter_color.npz"))
    params = [(z["order"][i].tolist(), float(z["factors"][i, 0]), float(z["factors"][i, 1]),
               float(z["factors"][i, 2]), float(z["hue"][i])) for i in range(z["frames"].shape[0])]
    return z, params


def test_oracle_jitter_matches_fixture():
    z, params = _jitter_fixture()
    h, w = (int(v) for v in z["size"])
    for i in range(z["frames"].shape[0]):
        r = FO.resize_bilinear_u8(z["frames"][i], h, w)
        if z["flip"][i]:
            r = r[:, ::-1]
        assert np.array_equal(FO.color_jitter_u8(np.ascontiguousarray(r), *params[i]), z["jittered_u8"][i])
    out = FO.transform_frames_jitter(z["frames"], h, w, z["mean"], z["std"], z["flip"], params)
    assert np.array_equal(out, z["out"])


def test_oracle_blend_matches_pillow_all_byte_pairs():
    Image = pytest.importorskip("PIL.Image")
    a = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 256, 1)
    b = np.ascontiguousarray(a.T)
    ia, ib = Image.fromarray(a), Image.fromarray(b)
    rng = np.random.default_rng(0)
    for alpha in list(rng.uniform(0.0, 2.5, 60)) + [0.0, 1.0, 0.5, 0.8, 1.2, float(np.float32(0.99999)), 1.0000001]:
        assert np.array_equal(FO.blend_u8(a, b, alpha), np.asarray(Image.blend(ia, ib, alpha))), alpha


def test_oracle_hsv_matches_pillow_all_triples():
    Image = pytest.importorskip("PIL.Image")
    g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for r0 in range(0, 256, 16):  # 16 red values per slab: 1 M triples
        slab = np.stack([np.broadcast_to(np.arange(r0, r0 + 16, dtype=np.uint8)[:, None, None], (16, 256, 256)),
                         np.broadcast_to(g, (16, 256, 256)), np.broadcast_to(b, (16, 256, 256))], -1)
        slab = np.ascontiguousarray(slab.reshape(16 * 256, 256, 3))
        assert np.array_equal(FO.rgb_to_hsv_u8(slab), np.asarray(Image.fromarray(slab, "RGB").convert("HSV")))
        assert np.array_equal(FO.hsv_to_rgb_u8(slab), np.asarray(Image.fromarray(slab, "HSV").convert("RGB")))


def test_oracle_jitter_matches_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    rng = np.random.default_rng(3)
    for it in range(24):
        a = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
        if it % 2:
            yy, xx = np.mgrid[0:37, 0:53]
            for c in range(3):
                a[:, :, c] = ((np.sin(xx / (3.0 + c) + it) + np.cos(yy / (5.0 - c))) * 63 + 128).clip(0, 255)
        order = rng.permutation(4).tolist()
        fb, fc, fs = (float(np.float32(v)) for v in rng.uniform(0.8, 1.2, 3))
        fh = float(np.float32(rng.uniform(-0.1, 0.1)))
        if it > 16:
            fb, fc, fs = (float(np.float32(v)) for v in rng.uniform(0.0, 2.5, 3))
            fh = float(np.float32(rng.uniform(-0.5, 0.5)))
        im = Image.fromarray(a)
        for op in order:
            if op == 0:
                im = ImageEnhance.Brightness(im).enhance(fb)
            elif op == 1:
                im = ImageEnhance.Contrast(im).enhance(fc)
            elif op == 2:
                im = ImageEnhance.Color(im).enhance(fs)
            else:
                hh, ss, vv = im.convert("HSV").split()
                np_h = np.array(hh, dtype=np.uint8)
                with np.errstate(over="ignore"):
                    np_h += np.array(fh * 255).astype(np.uint8)
                im = Image.merge("HSV", (Image.fromarray(np_h, "L"), ss, vv)).convert("RGB")
        assert np.array_equal(FO.color_jitter_u8(a, order, fb, fc, fs, fh), np.asarray(im)), (it, order)


def test_draw_jitter_ranges_and_switches():
    from video_classification_amd import frames as FR
    g = torch.Generator().manual_seed(0)
    order, factors, shift = FR.draw_jitter(64, 0.2, 0.2, 0.2, 0.1, generator=g)
    assert order.dtype == torch.uint8 and sorted(order[0].tolist()) == [0, 1, 2, 3]
    assert float(factors.min()) >= 0.8 - 1e-6 and float(factors.max()) <= 1.2 + 1e-6
    s = shift.to(torch.int64)
    assert bool(((s <= 25) | (s >= 231)).all())  # |hue| <= 0.1 -> int(h * 255) in -25 .. 25, wrapped
    order, factors, shift = FR.draw_jitter(8, brightness=0.3)
    assert bool(((order == 0) | (order == 255)).all()) and bool((order == 0).sum(1).eq(1).all())
    assert bool((factors[:, 1:] == 1).all()) and bool((shift == 0).all())
    assert FO.hue_shift_byte(-0.05) == (int(-0.05 * 255) & 255) == 244


# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_gpu_transform_matches_fixtures(cuda, path):
    from video_classification_amd import frames as FR
    z = _load(path)
    tf = FR.FrameTransform(tuple(int(v) for v in z["size"]), z["mean"].tolist(), z["std"].tolist(),
                           gray=bool(z["gray"]), device=cuda)
    out = tf(torch.from_numpy(z["frames"]).to(cuda), torch.from_numpy(z["flip"]).to(cuda))
    assert np.array_equal(out.cpu().numpy(), z["out"])  # bit-exact float32
    tf16 = FR.FrameTransform(tuple(int(v) for v in z["size"]), z["mean"].tolist(), z["std"].tolist(),
                             gray=bool(z["gray"]), device=cuda, dtype=torch.bfloat16)
    out16 = tf16(torch.from_numpy(z["frames"]).to(cuda), torch.from_numpy(z["flip"]).to(cuda))
    assert torch.equal(out16.cpu(), torch.from_numpy(z["out"]).to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize("n,hs,ws,size,gray", [
    (6, 256, 340, 224, False),    # jpegs_256 frames of UCF101 -> the colour models
    (6, 240, 320, 112, True),     # Conv3D
    (3, 256, 341, 224, False),    # odd row length: no frame but the first starts on a 16-byte boundary
    (2, 50, 70, 224, False),      # up-scaling
    (2, 1080, 1920, 224, False),  # 8.6x reduction: 19-tap windows, row blocks of more than 64 KB
    (2, 224, 224, 224, False),    # nothing to resize
    (1, 7, 5, (3, 9), True),
])
def test_gpu_transform_full_size(cuda, n, hs, ws, size, gray):
    from video_classification_amd import frames as FR
    rng = np.random.default_rng(n * 1000 + hs + ws)
    frames = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    flip = (rng.random(n) < 0.5).astype(np.uint8)
    flip[-1] = 1
    mean, std = ([0.5], [0.5]) if gray else (FR.IMAGENET_MEAN, FR.IMAGENET_STD)
    tf = FR.FrameTransform(size, mean, std, gray=gray, device=cuda)
    out = tf(torch.from_numpy(frames).to(cuda), torch.from_numpy(flip).to(cuda)).cpu().numpy()
    want = FO.transform_frames(frames, tf.hd, tf.wd, mean, std, flip, gray)
    assert out.shape == want.shape
    assert np.array_equal(out, want)
    # clip-shaped input [B, T, H, W, 3] -> [B, T, C, S, S], no flip
    if n % 2 == 0:
        out2 = tf(torch.from_numpy(frames).to(cuda).view(2, n // 2, hs, ws, 3)).cpu().numpy()
        want2 = FO.transform_frames(frames, tf.hd, tf.wd, mean, std, None, gray)
        assert out2.shape == (2, n // 2) + want2.shape[1:]
        assert np.array_equal(out2.reshape(want2.shape), want2)


@pytest.mark.gpu
def test_gpu_prefetcher_yields_reference_batches(cuda):
    from video_classification_amd import frames as FR
    rng = np.random.default_rng(11)
    batches = []
    for b in range(3):
        clips = [torch.from_numpy(rng.integers(0, 256, (4, 48, 64, 3), dtype=np.uint8)) for _ in range(3)]
        if b == 1:  # one clip of another frame size in the batch
            clips[2] = torch.from_numpy(rng.integers(0, 256, (4, 40, 40, 3), dtype=np.uint8))
        batches.append((clips, torch.tensor([[b], [b + 1], [b + 2]])))
    tf = FR.FrameTransform(32, device=cuda)
    gen = torch.Generator().manual_seed(5)
    got = list(FR.DevicePrefetcher(batches, tf, flip_p=0.5, generator=gen))
    gen = torch.Generator().manual_seed(5)
    assert len(got) == 3
    for (clips, target), (x, y) in zip(batches, got):
        flip = (torch.rand(12, generator=gen) < 0.5).numpy().reshape(3, 4)
        assert x.device.type == "cuda" and tuple(x.shape) == (3, 4, 3, 32, 32) and x.dtype == torch.float32
        assert torch.equal(y.cpu(), target)
        for i, cl in enumerate(clips):
            want = FO.transform_frames(cl.numpy(), 32, 32, FR.IMAGENET_MEAN, FR.IMAGENET_STD, flip[i])
            assert np.array_equal(x[i].cpu().numpy(), want)
    # gray pipeline hands out Conv3D's [B, 1, T, S, S]
    tfg = FR.FrameTransform(16, [0.5], [0.5], gray=True, device=cuda)
    stacked = [(torch.stack(c[0][:2]), c[1][:2]) for c in batches[:1]]
    (x, y), = list(FR.DevicePrefetcher(stacked, tfg))
    assert tuple(x.shape) == (2, 1, 4, 16, 16)
    want = FO.transform_frames(stacked[0][0][1].numpy(), 16, 16, [0.5], [0.5], None, True)
    assert np.array_equal(x[1, 0].cpu().numpy(), want[:, 0])


@pytest.mark.gpu
def test_gpu_jitter_matches_fixture(cuda):
    from video_classification_amd import frames as FR
    z, _ = _jitter_fixture()
    tf = FR.FrameTransform(tuple(int(v) for v in z["size"]), z["mean"].tolist(), z["std"].tolist(), device=cuda)
    shift = torch.tensor([FO.hue_shift_byte(float(h)) for h in z["hue"]], dtype=torch.uint8)
    out = tf(torch.from_numpy(z["frames"]).to(cuda), torch.from_numpy(z["flip"]).to(cuda),
             jitter_params=(torch.from_numpy(z["order"]), torch.from_numpy(z["factors"]), shift))
    assert np.array_equal(out.cpu().numpy(), z["out"])


@pytest.mark.gpu
def test_gpu_hsv_round_trip_all_triples(cuda):
    """Every RGB triple through the hue operation of vc_frames_jitter (C ABI called directly on planar uint8
    frames) against the oracle, for a zero and a non-zero hue shift."""
    from video_classification_amd import _lib as L
    r, g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8),
                          np.arange(256, dtype=np.uint8), indexing="ij")
    planes = torch.from_numpy(np.stack([r.reshape(16, -1), g.reshape(16, -1), b.reshape(16, -1)], 1).copy())
    n, hw = 16, 1 << 20
    mean = torch.zeros(3, device=cuda)
    std = torch.ones(3, device=cuda)
    order = torch.tensor([[3, 255, 255, 255]] * n, dtype=torch.uint8, device=cuda)
    factors = torch.ones(n, 3, device=cuda)
    rgb = np.stack([r.reshape(-1), g.reshape(-1), b.reshape(-1)], -1)
    for sh in (0, 37):
        p = planes.to(cuda)
        shift = torch.full((n,), sh, dtype=torch.uint8, device=cuda)
        out = torch.empty(n, 3, hw, device=cuda)
        L.call("vc_frames_jitter", L.ptr(p), n, 1024, 1024, L.ptr(order), L.ptr(factors), L.ptr(shift), L.ptr(mean),
               L.ptr(std), L.ptr(out), 0, L.stream())
        got = out.cpu().numpy().transpose(0, 2, 1).reshape(-1, 3)
        hsv = FO.rgb_to_hsv_u8(rgb)
        hsv[:, 0] = (hsv[:, 0].astype(np.int32) + sh) & 255
        want = FO.hsv_to_rgb_u8(hsv).astype(np.float32) / np.float32(255)
        assert np.array_equal(got, want), sh


@pytest.mark.gpu
def test_gpu_jitter_full_size_random_chains(cuda):
    from video_classification_amd import frames as FR
    rng = np.random.default_rng(21)
    n, hs, ws, S = 10, 256, 340, 224
    frames = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:hs, 0:ws]
    for i in range(0, n, 2):
        for c in range(3):
            frames[i, :, :, c] = ((np.sin(xx / (9.0 + c) + i) + np.cos(yy / (15.0 - c))) * 63 + 128).clip(0, 255)
    flip = (rng.random(n) < 0.5).astype(np.uint8)
    order = np.stack([rng.permutation(4) for _ in range(n)]).astype(np.uint8)
    order[1, order[1] == 1] = 255   # no contrast: a single sweep
    order[2] = (1, 0, 2, 3)         # contrast first: the first sweep only sums
    order[3] = (0, 2, 3, 1)         # contrast last
    order[4] = (255, 255, 255, 255)
    factors = rng.uniform(0.8, 1.2, (n, 3)).astype(np.float32)
    factors[5] = (0.0, 2.5, 1.9)
    factors[6] = (1.0, 1.0, 0.0)
    hue = rng.uniform(-0.1, 0.1, n).astype(np.float32)
    hue[7] = 0.5
    shift = torch.tensor([FO.hue_shift_byte(float(h)) for h in hue], dtype=torch.uint8)
    tf = FR.FrameTransform(S, device=cuda)
    out = tf(torch.from_numpy(frames).to(cuda), torch.from_numpy(flip).to(cuda),
             jitter_params=(torch.from_numpy(order), torch.from_numpy(factors), shift)).cpu().numpy()
    params = [([int(o) for o in order[i]], float(factors[i, 0]), float(factors[i, 1]), float(factors[i, 2]),
               float(hue[i])) for i in range(n)]
    want = FO.transform_frames_jitter(frames, S, S, FR.IMAGENET_MEAN, FR.IMAGENET_STD, flip, params)
    for i in range(n):
        assert np.array_equal(out[i], want[i]), (i, params[i])
    # parameters drawn inside the transform: reproducible from the generator, and a different clip from the plain one
    tfj = FR.FrameTransform(S, device=cuda, jitter=dict(brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1),
                            generator=torch.Generator().manual_seed(9))
    a = tfj(torch.from_numpy(frames[:4]).to(cuda))
    drawn = FR.draw_jitter(4, 0.2, 0.2, 0.2, 0.1, generator=torch.Generator().manual_seed(9))
    b2 = tf(torch.from_numpy(frames[:4]).to(cuda), jitter_params=drawn)
    assert torch.equal(a, b2) and not torch.equal(a, tf(torch.from_numpy(frames[:4]).to(cuda)))


@pytest.mark.gpu
def test_gpu_epoch_from_jpeg_folders(cuda, tmp_path):
    """The whole caller chain of the reference's training loop on the device-side pipeline: JPEG folders ->
    RawFrameDataset -> DataLoader(raw_collate) -> DevicePrefetcher (resize, flip off) -> train_epoch /
    validate_epoch of a small CRNN, against the CPU oracle fed with the host-side transform of the same files
    (Pillow resize + ToTensor + Normalize, what Dataset_CRNN yields: data_loaders.py:88-108)."""
    Image = pytest.importorskip("PIL.Image")
    import types

    import torch.nn as nn
    from torch.utils import data

    from oracle import torch_oracle as O
    from video_classification_amd import frames as FR
    from video_classification_amd import modules as M
    from video_classification_amd import trainer as TR

    rng = np.random.default_rng(4)
    folders, labels = [], []
    yy, xx = np.mgrid[0:72, 0:96]
    for v in range(4):
        name = f"v_Class{v}_g01_c01"
        os.makedirs(tmp_path / name)
        for i in range(1, 5):
            img = np.stack([((np.sin(xx / (5.0 + c + v) + i) + np.cos(yy / (7.0 + v))) * 60 + 128) for c in range(3)],
                           -1).clip(0, 255).astype(np.uint8)
            Image.fromarray(img).save(str(tmp_path / name / f"frame{i:06d}.jpg"), quality=92)
        folders.append(name)
        labels.append(v % 3)
    frames_sel, S = [1, 2, 3, 4], 64

    ds = FR.RawFrameDataset(str(tmp_path), folders, labels, frames_sel)
    loader = data.DataLoader(ds, batch_size=2, shuffle=False, collate_fn=FR.raw_collate, num_workers=0)
    tf = FR.FrameTransform(S, device=cuda)
    dev_loader = FR.DevicePrefetcher(loader, tf, flip_p=0.0)

    def host_clip(folder):  # the reference's Dataset_CRNN.read_images with the train.py transform, flip off
        out = []
        for i in frames_sel:
            im = Image.open(str(tmp_path / folder / f"frame{i:06d}.jpg")).resize((S, S), Image.BILINEAR)
            t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).contiguous().float().div(255)
            m = torch.tensor(FR.IMAGENET_MEAN)[:, None, None]
            s = torch.tensor(FR.IMAGENET_STD)[:, None, None]
            out.append(t.sub_(m).div_(s))
        return torch.stack(out)

    host_batches = [(torch.stack([host_clip(f) for f in folders[b:b + 2]]),
                     torch.tensor([[labels[b]], [labels[b + 1]]])) for b in (0, 2)]
    got = list(dev_loader)
    assert len(got) == 2 == len(dev_loader)
    for (hx, hy), (dx, dy) in zip(host_batches, got):
        assert torch.equal(dx.cpu(), hx) and torch.equal(dy.cpu(), hy)  # the clips are bit-identical

    cfg = types.SimpleNamespace(img_x=S, img_y=S, CNN_fc_hidden1=64, CNN_fc_hidden2=48, CNN_embed_dim=32,
                                dropout_p=0.0, use_pretrained=False, RNN_hidden_layers=2, RNN_hidden_nodes=32,
                                RNN_FC_dim=16, num_classes=3)
    torch.manual_seed(0)
    ref = O.CRNNModel(cfg)
    mine = M.CRNNModel(cfg)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda).set_compute_dtype(torch.float32)
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-4)
    opt_m = TR.get_optimizer(mine, "adam", lr=1e-3, weight_decay=1e-4)
    crit = nn.CrossEntropyLoss()
    losses = [O.train_step(ref, x, y, crit, opt_r)[0] for x, y in host_batches]
    loss_m, acc_m = TR.train_epoch(mine, dev_loader, TR.FusedCrossEntropyLoss(), opt_m, cuda, 1)
    assert loss_m == pytest.approx(sum(losses) / len(losses), rel=2e-4)
    assert 0.0 <= acc_m <= 100.0
    vloss, vacc = TR.validate_epoch(mine, dev_loader, TR.FusedCrossEntropyLoss(), cuda, 1)
    ref.eval()
    with torch.no_grad():
        want = sum(float(crit(ref(x), y.squeeze(1))) for x, y in host_batches) / 2
    assert vloss == pytest.approx(want, rel=2e-3)


@pytest.mark.gpu
def test_gpu_transform_random_geometries(cuda):
    """Seeded sweep over source / target sizes (odd widths take the scalar store path, tiny frames have windows
    clipped at both borders, strong reductions use long windows and small row blocks), colour and luma, both output
    types, against the oracle bit for bit."""
    from video_classification_amd import frames as FR
    rng = np.random.default_rng(77)
    for case in range(28):
        hs, ws = int(rng.integers(1, 200)), int(rng.integers(1, 260))
        hd, wd = int(rng.integers(1, 120)), int(rng.integers(1, 130))
        if case % 7 == 0:
            hs, ws = int(rng.integers(300, 700)), int(rng.integers(300, 900))  # 3x .. 40x reduction
        gray = bool(case % 3 == 0)
        n = int(rng.integers(1, 4))
        frames = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
        flip = (rng.random(n) < 0.5).astype(np.uint8)
        mean, std = ([0.5], [0.5]) if gray else (FR.IMAGENET_MEAN, FR.IMAGENET_STD)
        tf = FR.FrameTransform((hd, wd), mean, std, gray=gray, device=cuda)
        out = tf(torch.from_numpy(frames).to(cuda), torch.from_numpy(flip).to(cuda))
        want = FO.transform_frames(frames, hd, wd, mean, std, flip, gray)
        assert np.array_equal(out.cpu().numpy(), want), (case, hs, ws, hd, wd, gray)
        if case % 4 == 1:
            tf16 = FR.FrameTransform((hd, wd), mean, std, gray=gray, device=cuda, dtype=torch.bfloat16)
            out16 = tf16(torch.from_numpy(frames).to(cuda), torch.from_numpy(flip).to(cuda))
            assert torch.equal(out16.cpu(), torch.from_numpy(want).to(torch.bfloat16)), (case, hs, ws, hd, wd)
