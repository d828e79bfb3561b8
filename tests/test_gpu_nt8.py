"""The 256 x 256-tile, 8-wave form of the NT kernel (csrc/vc_igemm8.hip) against the 128-wide persistent kernel and
against the plain-FMA kernel, through the C ABI.  VC_NT8=1 sends every geometry the wide kernel can run to it, VC_NT8=0
none (the library reads the variable at every call).  Both kernels multiply the same bf16 operands with the same MFMA
instruction in the same k order and round the fp32 sums once: stored bf16 results are compared bit for bit.  The
per-slab BatchNorm sums are added in a different lane order by the two epilogues: 1e-5 of the largest magnitude.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


class _Wide:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("VC_NT8")
        os.environ["VC_NT8"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["VC_NT8"]
        else:
            os.environ["VC_NT8"] = self.old


SHAPES = [
    # N, H, W, Ci, Co, k: ragged rows and channels / a channel tail on one tap / two tiles deep in k / many tiles
    (3, 13, 11, 192, 320, 3),
    (5, 9, 7, 136, 200, 1),
    (20, 7, 7, 512, 512, 3),
    (16, 14, 14, 256, 1024, 1),
    (9, 14, 14, 1024, 256, 1),
]


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_forward_and_data_gradient_match_the_128_wide_kernel(cuda, shape):
    from video_classification_amd import _lib as L
    N, H, W, Ci, Co, k = shape
    g = torch.Generator().manual_seed(31)
    p, taps, M = k // 2, k * k, N * H * W
    x = torch.randn(M, Ci, generator=g).to(cuda, torch.bfloat16)
    dy = torch.randn(M, Co, generator=g).to(cuda, torch.bfloat16)
    wf = (torch.randn(Co, taps, Ci, generator=g) * 0.1).to(cuda, torch.bfloat16)
    wt = wf.permute(2, 1, 0).contiguous()
    bias = torch.randn(Co, generator=g).to(cuda)
    d = L.int_array([N, 1, H, W, Ci, 1, H, W, Co, 1, k, k, 1, 1, 1, 0, p, p])
    slabs = (M + 63) // 64
    out = {}
    for wide in (0, 1):
        with _Wide(wide):
            y = torch.full((M, Co), float("nan"), device=cuda, dtype=torch.bfloat16)
            yr = torch.full_like(y, float("nan"))
            ys = torch.full_like(y, float("nan"))
            part = torch.full((slabs, 2, Co), float("nan"), device=cuda)
            dx = torch.full((M, Ci), float("nan"), device=cuda, dtype=torch.bfloat16)
            dxa = torch.ones(M, Ci, device=cuda, dtype=torch.bfloat16)
            L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y), 0, 0, 0, 0, L.stream())
            L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(bias), L.ptr(yr), 0, 1, 0, 0, L.stream())
            if Co % 8 == 0 and L.query("vc_conv_fwd_stats_supported", d, 1, 0, 0) == 1:
                L.call("vc_conv_fwd_stats", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(ys), 0, 0, L.ptr(part),
                       L.stream())
            L.call("vc_conv_dgrad", d, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx), 0, 0, 0, L.stream())
            L.call("vc_conv_dgrad", d, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dxa), 0, 0, 1, L.stream())  # accumulate
            torch.cuda.synchronize()
            out[wide] = (y, yr, ys, part, dx, dxa)
    for a, b, what in zip(out[0], out[1], ("y", "relu(y + bias)", "y with sums", "slab sums", "dx", "dx accumulated")):
        if what == "slab sums":
            if not torch.isnan(a).any():
                assert not torch.isnan(b).any()
                assert _rel(b, a) < 1e-5, what
        elif not torch.isnan(a.float()).any():
            assert torch.equal(a, b), what
    # and against the plain-FMA kernel on the same data (fp32 result, rounded once here)
    yn = torch.empty(M, Co, device=cuda, dtype=torch.float32)
    L.call("vc_conv_fwd_naive_bf16", d, 0, L.ptr(x), L.ptr(wf), L.ptr(None), L.ptr(yn), 0, 1, L.stream())
    torch.cuda.synchronize()
    assert _rel(out[1][0].float(), yn) < 8e-3  # one bf16 rounding of a sum formed in a different order


@pytest.mark.parametrize("shape,beta", [((4, 8, 8, 64, 256), 1), ((3, 7, 9, 128, 512), 1), ((6, 14, 14, 2048, 512), 0)],
                         ids=["64_256_acc", "128_512_ragged_acc", "2048_512_write"])
def test_data_gradient_with_recorded_relu_decisions(cuda, shape, beta):
    """vc_conv_dgrad_bn_tail on both tile widths: identical masked gradients, slab sums to summation order."""
    from video_classification_amd import _lib as L
    N, H, W, Cmid, Cout = shape
    g = torch.Generator().manual_seed(32)
    M = N * H * W
    dy = torch.randn(M, Cmid, generator=g).to(cuda, torch.bfloat16)
    wt = (torch.randn(Cout, 1, Cmid, generator=g) * 0.1).to(cuda, torch.bfloat16)
    old = torch.randn(M, Cout, generator=g).to(cuda, torch.bfloat16)
    y = (torch.randn(M, Cout, generator=g) * 2 + 0.3).to(cuda, torch.bfloat16)
    bits = torch.randint(0, 256, (M, Cout // 8), generator=g, dtype=torch.uint8).to(cuda)
    d = L.int_array([N, 1, H, W, Cout, 1, H, W, Cmid, 1, 1, 1, 1, 1, 1, 0, 0, 0])
    assert L.query("vc_conv_dgrad_bn_tail_supported", d, 1) == 1
    slabs = (M + 63) // 64
    res = {}
    for wide in (0, 1):
        with _Wide(wide):
            dx = old.clone()
            part = torch.full((slabs, 2, Cout), float("nan"), device=cuda)
            L.call("vc_conv_dgrad_bn_tail", d, 1, L.ptr(dy), L.ptr(wt), L.ptr(dx), beta, L.ptr(y), L.ptr(bits),
                   L.ptr(part), L.stream())
            torch.cuda.synchronize()
            res[wide] = (dx, part)
    assert torch.equal(res[0][0], res[1][0])
    assert not torch.isnan(res[1][1]).any()
    assert _rel(res[1][1], res[0][1]) < 1e-5


@pytest.mark.parametrize("shape", [(2, 16, 16, 256, 128, 1, 1), (2, 16, 16, 192, 192, 3, 1), (128, 7, 7, 512, 512, 3, 2)],
                         ids=["1x1_256", "3x3_192", "3x3_512_two_groups"])
def test_data_gradient_with_recomputed_relu_decisions(cuda, shape):
    """vc_conv_dgrad_bn_pre on both tile widths: identical masked gradients, slab sums to summation order."""
    from video_classification_amd import _lib as L
    N, H, W, Cin, Cmid, k, G = shape
    g = torch.Generator().manual_seed(33)
    M = N * H * W
    if M % (64 * G):
        pytest.skip("rows per group must be a multiple of 64")
    rpg = M // G
    dy = torch.randn(M, Cmid, generator=g).to(cuda, torch.bfloat16)
    wt = (torch.randn(Cin, k * k, Cmid, generator=g) * 0.1).to(cuda, torch.bfloat16)
    y = (torch.randn(M, Cin, generator=g) * 2 + 0.3).to(cuda, torch.bfloat16)
    stat = torch.randn(4, G, Cin, generator=g).to(cuda)
    stat[1] = stat[1].abs() + 0.5
    d = L.int_array([N, 1, H, W, Cin, 1, H, W, Cmid, 1, k, k, 1, 1, 1, 0, k // 2, k // 2])
    assert L.query("vc_conv_dgrad_bn_pre_supported", d, 1, rpg) == 1
    res = {}
    for wide in (0, 1):
        with _Wide(wide):
            dx = torch.full((M, Cin), float("nan"), device=cuda, dtype=torch.bfloat16)
            part = torch.full((M // 64, 2, Cin), float("nan"), device=cuda)
            L.call("vc_conv_dgrad_bn_pre", d, 1, L.ptr(dy), L.ptr(wt), L.ptr(dx), L.ptr(y), L.ptr(stat), rpg, L.ptr(part),
                   L.stream())
            torch.cuda.synchronize()
            res[wide] = (dx, part)
    assert not torch.isnan(res[1][0].float()).any()
    assert torch.equal(res[0][0], res[1][0])
    assert _rel(res[1][1], res[0][1]) < 1e-5


def test_wide_kernel_repeats_bit_identically(cuda):
    """The persistent walk (several tiles per workgroup, requests running through the tile boundaries) is free of races:
    20 launches of a layer with 392 tiles on 256 workgroups give one result."""
    from video_classification_amd import _lib as L
    N, H, Ci, Co, k = 512, 14, 256, 256, 3
    g = torch.Generator(device="cuda").manual_seed(34)
    M = N * H * H
    x = torch.randn(M, Ci, device=cuda, generator=g).to(torch.bfloat16)
    wf = (torch.randn(Co, k * k, Ci, device=cuda, generator=g) * 0.1).to(torch.bfloat16)
    d = L.int_array([N, 1, H, H, Ci, 1, H, H, Co, 1, k, k, 1, 1, 1, 0, 1, 1])
    with _Wide(0):
        ref = torch.empty(M, Co, device=cuda, dtype=torch.bfloat16)
        L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(ref), 0, 0, 0, 0, L.stream())
    with _Wide(1):
        for _ in range(20):
            y = torch.full((M, Co), float("nan"), device=cuda, dtype=torch.bfloat16)
            L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y), 0, 0, 0, 0, L.stream())
            torch.cuda.synchronize()
            assert torch.equal(y, ref)


class _WideW(_Wide):
    def __enter__(self):
        self.old = os.environ.get("VC_WGRAD8")
        os.environ["VC_WGRAD8"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["VC_WGRAD8"]
        else:
            os.environ["VC_WGRAD8"] = self.old


WSHAPES = [
    # N, H, W, Ci, Co, k, s: ragged everything / channel tail, one tap / strided 3x3 / many slices, then the narrow tiles:
    # 64 x 256 (3x3, 64 channels) / 256 x 64 / 64 x 256 (1x1) / 128 x 256 / 256 x 128
    (3, 13, 11, 192, 320, 3, 1),
    (5, 9, 7, 136, 200, 1, 1),
    (8, 28, 28, 256, 256, 3, 2),
    (64, 14, 14, 256, 1024, 1, 1),
    (4, 56, 56, 64, 64, 3, 1),
    (3, 31, 29, 64, 256, 1, 1),
    (3, 31, 29, 256, 64, 1, 1),
    (5, 28, 28, 128, 128, 3, 1),
    (7, 28, 27, 128, 512, 1, 1),
    # linear layers of a Swin block (bias gradient inside the kernel): 96 -> 384, 384 -> 96, 192 -> 768
    (2, 56, 56, 96, 384, 1, 1),
    (2, 56, 56, 384, 96, 1, 1),
    (4, 28, 28, 192, 768, 1, 1),
]


@pytest.mark.parametrize("shape", WSHAPES, ids=["x".join(map(str, s)) for s in WSHAPES])
@pytest.mark.parametrize("det", [False, True], ids=["atomics", "ordered"])
def test_weight_gradient_on_wide_tiles(cuda, shape, det):
    """The 256 x 256-tile weight-gradient kernel (csrc/vc_wgrad8.hip) against the plain-FMA kernel on the same bf16 data
    (fp32 sums in a different order: 2e-4 of the largest magnitude, the bound of the 128-wide kernel's test), with the
    bias gradient, and in deterministic mode (slices added in order): two runs bit-identical."""
    from video_classification_amd import _lib as L
    from video_classification_amd import ops
    N, H, W, Ci, Co, k, s = shape
    g = torch.Generator().manual_seed(41)
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N * H * W, Ci, generator=g).to(cuda, torch.bfloat16)
    dy = torch.randn(N * Ho * Wo, Co, generator=g).to(cuda, torch.bfloat16)
    d = L.int_array([N, 1, H, W, Ci, 1, Ho, Wo, Co, 1, k, k, 1, s, s, 0, p, p])
    dwn = torch.zeros(Co, k * k, Ci, device=cuda)
    L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dwn), L.ptr(None), 1, L.stream())
    torch.cuda.synchronize()
    runs = []
    with _WideW(1):
        ops.set_deterministic(det)
        try:
            for _ in range(2):
                dw = torch.zeros(Co, k * k, Ci, device=cuda)
                db = torch.zeros(Co, device=cuda)
                L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dw), L.ptr(db), 0, L.stream())
                torch.cuda.synchronize()
                runs.append((dw, db))
        finally:
            ops.set_deterministic(False)
    assert _rel(runs[0][0], dwn) < 2e-4
    assert _rel(runs[0][1], dy.float().sum(0)) < 2e-4
    if det:
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    # accumulation into a non-zero dw (and bias gradient)
    with _WideW(1):
        dw = torch.ones(Co, k * k, Ci, device=cuda)
        L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dw), L.ptr(None), 0, L.stream())
        dw2 = torch.ones(Co, k * k, Ci, device=cuda)
        db2 = torch.ones(Co, device=cuda)
        L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dw2), L.ptr(db2), 0, L.stream())
        torch.cuda.synchronize()
    assert _rel(dw - 1, dwn) < 2e-4
    assert _rel(dw2 - 1, dwn) < 2e-4
    assert _rel(db2 - 1, dy.float().sum(0)) < 2e-4


def test_wide_kernels_on_a_conv3d_geometry(cuda):
    """Three filter axes (27 taps, the last Conv3D layer's shape at a small clip): forward and data gradient on the wide
    tile bit-identical to the 128-wide kernel, weight gradient on the wide tile against the plain-FMA kernel."""
    from video_classification_amd import _lib as L
    N, D, H, W, Ci, Co = 2, 4, 14, 12, 128, 256
    g = torch.Generator().manual_seed(51)
    M = N * D * H * W
    x = torch.randn(M, Ci, generator=g).to(cuda, torch.bfloat16)
    dy = torch.randn(M, Co, generator=g).to(cuda, torch.bfloat16)
    wf = (torch.randn(Co, 27, Ci, generator=g) * 0.05).to(cuda, torch.bfloat16)
    wt = wf.permute(2, 1, 0).contiguous()
    d = L.int_array([N, D, H, W, Ci, D, H, W, Co, 3, 3, 3, 1, 1, 1, 1, 1, 1])
    res = {}
    for wide in (0, 1):
        with _Wide(wide):
            y = torch.full((M, Co), float("nan"), device=cuda, dtype=torch.bfloat16)
            dx = torch.full((M, Ci), float("nan"), device=cuda, dtype=torch.bfloat16)
            L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y), 0, 0, 0, 0, L.stream())
            L.call("vc_conv_dgrad", d, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx), 0, 0, 0, L.stream())
            torch.cuda.synchronize()
            res[wide] = (y, dx)
    assert not torch.isnan(res[1][0].float()).any() and not torch.isnan(res[1][1].float()).any()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    yn = torch.empty(M, Co, device=cuda, dtype=torch.float32)
    L.call("vc_conv_fwd_naive_bf16", d, 0, L.ptr(x), L.ptr(wf), L.ptr(None), L.ptr(yn), 0, 1, L.stream())
    dwn = torch.zeros(Co, 27, Ci, device=cuda)
    L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dwn), L.ptr(None), 1, L.stream())
    torch.cuda.synchronize()
    assert _rel(res[1][0].float(), yn) < 8e-3
    for il in ("1", "0"):  # interleaved and contiguous row slices
        os.environ["VC_WGRAD8_IL"] = il
        try:
            with _WideW(1):
                dw = torch.zeros(Co, 27, Ci, device=cuda)
                L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dw), L.ptr(None), 0, L.stream())
                torch.cuda.synchronize()
        finally:
            del os.environ["VC_WGRAD8_IL"]
        assert _rel(dw, dwn) < 2e-4, il
