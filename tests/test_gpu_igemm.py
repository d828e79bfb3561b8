"""GPU parity of the implicit-GEMM kernels (conv2d / conv3d / linear; forward, data gradient,
weight gradient) against torch's fp32 CPU convolution on the same inputs, through the C ABI.

bf16 path with fp32 output: inputs are exactly representable in bf16 and the MFMA accumulates
in fp32, so the tolerance is fp32-accumulation-order only (2e-4 relative to the output scale).
fp32 path: 1e-5 relative.
"""
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # name, N, D, H, W, Ci, Co, k(d,h,w), s, p
    ("linear_small", 100, 1, 1, 1, 72, 40, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("linear_big", 300, 1, 1, 1, 512, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("linear_101", 37, 1, 1, 1, 256, 104, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("c2d_3x3_s1_p1", 3, 1, 13, 11, 16, 24, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("c2d_3x3_s2_p0", 2, 1, 27, 27, 32, 64, (1, 3, 3), (1, 2, 2), (0, 0, 0)),
    ("c2d_5x5_s2_p0", 2, 1, 32, 30, 8, 32, (1, 5, 5), (1, 2, 2), (0, 0, 0)),
    ("c2d_1x1_s2_p0", 2, 1, 14, 14, 64, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("c2d_7x7_s2_p3", 2, 1, 32, 32, 8, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
    ("c2d_3x3_s2_p1", 2, 1, 28, 28, 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("c2d_wide", 4, 1, 14, 14, 256, 264, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("c2d_3x3_64_64", 3, 1, 14, 13, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("c2d_1x1_256_64", 2, 1, 9, 11, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("c2d_1x1_64_256", 5, 1, 12, 12, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("c2d_3x3_s2_64_40", 2, 1, 17, 17, 64, 40, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("c3d_3x3x3_p1", 2, 4, 10, 9, 8, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("c3d_3x3x3_p1_b", 1, 5, 7, 12, 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
]


def _desc(N, D, H, W, Ci, Co, k, s, p):
    Do = (D + 2 * p[0] - k[0]) // s[0] + 1
    Ho = (H + 2 * p[1] - k[1]) // s[1] + 1
    Wo = (W + 2 * p[2] - k[2]) // s[2] + 1
    return [N, D, H, W, Ci, Do, Ho, Wo, Co, *k, *s, *p], (Do, Ho, Wo)


def _to_ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_conv_fwd_dgrad_wgrad(cuda, case, dt):
    from video_classification_amd import _lib as L
    name, N, D, H, W, Ci, Co, k, s, p = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn(N, Ci, D, H, W, generator=g)
    w = torch.randn(Co, Ci, *k, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    desc, (Do, Ho, Wo) = _desc(N, D, H, W, Ci, Co, k, s, p)
    dy = torch.randn(N, Co, Do, Ho, Wo, generator=g)
    if dt == torch.bfloat16:
        dy = dy.bfloat16().float()

    # CPU fp32 reference
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, b, stride=s, padding=p)
    yr.backward(dy)

    taps = k[0] * k[1] * k[2]
    code = L.dtype_code(dt)
    xd = _to_ndhwc(x).to(cuda, dt)
    dyd = _to_ndhwc(dy).to(cuda, dt)
    wd = w.reshape(Co, Ci, taps).contiguous().to(cuda)
    bd = b.to(cuda)
    wf = torch.empty(Co, taps, Ci, device=cuda, dtype=dt)
    wt = torch.empty(Ci, taps, Co, device=cuda, dtype=dt)
    L.call("vc_pack_weight", code, L.ptr(wd), L.ptr(wf), L.ptr(wt), Co, Ci, taps, Ci, Co, L.stream())
    d = L.int_array(desc)

    tol = 2e-4 if dt == torch.bfloat16 else 1e-5
    # forward, fp32 output
    y = torch.empty(N, Do, Ho, Wo, Co, device=cuda, dtype=torch.float32)
    L.call("vc_conv_fwd", d, code, L.ptr(xd), 0, L.ptr(wf), L.ptr(bd), L.ptr(y), 0, 0, 1, 0, L.stream())
    torch.cuda.synchronize()
    assert _rel(y.cpu(), _to_ndhwc(yr.detach())) < tol

    # forward, compute-type output with ReLU
    y2 = torch.empty(N, Do, Ho, Wo, Co, device=cuda, dtype=dt)
    L.call("vc_conv_fwd", d, code, L.ptr(xd), 0, L.ptr(wf), L.ptr(bd), L.ptr(y2), 0, 1, int(dt == torch.float32), 0,
           L.stream())
    torch.cuda.synchronize()
    assert _rel(y2.float().cpu(), _to_ndhwc(yr.detach()).relu()) < (1e-2 if dt == torch.bfloat16 else tol)

    # data gradient
    dx = torch.empty(N, D, H, W, Ci, device=cuda, dtype=torch.float32)
    L.call("vc_conv_dgrad", d, code, L.ptr(dyd), 0, L.ptr(wt), L.ptr(dx), 0, 1, 0, L.stream())
    torch.cuda.synchronize()
    assert _rel(dx.cpu(), _to_ndhwc(xr.grad)) < tol
    # accumulate into what dx holds (beta = 1), compute-type output: dx0 + dgrad
    dx0 = torch.randn(N, D, H, W, Ci, generator=g).to(cuda, dt)
    dxa = dx0.clone()
    L.call("vc_conv_dgrad", d, code, L.ptr(dyd), 0, L.ptr(wt), L.ptr(dxa), 0, int(dt == torch.float32), 1, L.stream())
    torch.cuda.synchronize()
    assert _rel(dxa.float().cpu(), _to_ndhwc(xr.grad) + dx0.float().cpu()) < (1e-2 if dt == torch.bfloat16 else tol)

    # weight gradient (MFMA path for bf16, naive path for fp32)
    dwp = torch.zeros(Co, taps, Ci, device=cuda, dtype=torch.float32)
    db = torch.zeros(Co, device=cuda, dtype=torch.float32)  # bias gradient from the same pass over dy
    L.call("vc_conv_wgrad", d, code, L.ptr(xd), 0, L.ptr(dyd), 0, L.ptr(dwp), L.ptr(db), 0, L.stream())
    dw = torch.empty(Co, Ci, taps, device=cuda, dtype=torch.float32)
    L.call("vc_unpack_wgrad", L.ptr(dwp), L.ptr(dw), Co, Ci, taps, Ci, 0, L.stream())
    torch.cuda.synchronize()
    assert _rel(dw.cpu(), wr.grad.reshape(Co, Ci, taps)) < tol
    assert _rel(db.cpu(), dy.sum(dim=(0, 2, 3, 4))) < tol

    if dt == torch.bfloat16:
        # on-device cross-check: naive kernels on the same bf16 data
        yn = torch.empty_like(y)
        L.call("vc_conv_fwd_naive_bf16", d, 0, L.ptr(xd), L.ptr(wf), L.ptr(bd), L.ptr(yn), 0, 1, L.stream())
        dwn = torch.zeros_like(dwp)
        L.call("vc_conv_wgrad", d, code, L.ptr(xd), 0, L.ptr(dyd), 0, L.ptr(dwn), L.ptr(None), 1, L.stream())
        torch.cuda.synchronize()
        assert _rel(y, yn) < tol
        assert _rel(dwp, dwn) < tol


def test_bad_shape_is_reported(cuda):
    from video_classification_amd import _lib as L
    desc = L.int_array([1, 1, 8, 8, 8, 1, 5, 5, 8, 1, 3, 3, 1, 1, 1, 0, 0, 0])  # Ho should be 6
    t = torch.zeros(8, device=cuda)
    with pytest.raises(RuntimeError, match="output extent"):
        L.call("vc_conv_fwd", desc, 1, L.ptr(t), 0, L.ptr(t), L.ptr(None), L.ptr(t), 0, 0, 0, 0, L.stream())


@pytest.mark.parametrize("shape", [(5, 12, 12, 64, 256, 1), (3, 14, 13, 64, 64, 3), (2, 9, 11, 256, 40, 1)],
                         ids=["1x1_64_256", "3x3_64_64", "1x1_256_40"])
def test_conv_fwd_with_bn_slab_sums(cuda, shape):
    """vc_conv_fwd_stats: same y as vc_conv_fwd (bit-exact) and slab sums equal to sums over the stored y."""
    from video_classification_amd import _lib as L
    N, H, W, Ci, Co, k = shape
    g = torch.Generator().manual_seed(7)
    p = k // 2
    cop = (Co + 7) // 8 * 8
    x = torch.randn(N, 1, H, W, Ci, generator=g).to(cuda, torch.bfloat16)
    wf = (torch.randn(cop, k * k, Ci, generator=g) * 0.1).to(cuda, torch.bfloat16)
    wf[Co:] = 0
    desc = L.int_array([N, 1, H, W, Ci, 1, H, W, cop, 1, k, k, 1, 1, 1, 0, p, p])
    assert L.query("vc_conv_fwd_stats_supported", desc, 1, 0, 0) == 1
    M = N * H * W
    slabs = (M + 63) // 64
    y0 = torch.empty(M, cop, device=cuda, dtype=torch.bfloat16)
    y1 = torch.empty_like(y0)
    part = torch.full((slabs, 2, cop), float("nan"), device=cuda)
    L.call("vc_conv_fwd", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y0), 0, 0, 0, 0, L.stream())
    L.call("vc_conv_fwd_stats", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y1), 0, 0, L.ptr(part), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    yf = y1.float().cpu()
    pad = slabs * 64 - M
    yp = torch.cat([yf, yf.new_zeros(pad, cop)]).view(slabs, 64, cop)
    ref = torch.stack([yp.sum(1), (yp * yp).sum(1)], 1)
    assert not torch.isnan(part).any()
    assert _rel(part.cpu(), ref) < 1e-5
    # group sums from the partials
    sums = torch.zeros(2, 1, cop, device=cuda)
    if M % 64 == 0:
        L.call("vc_bn_stats_from_partial", L.ptr(part), L.ptr(sums[0]), L.ptr(sums[1]), 1, slabs, cop, L.stream())
        torch.cuda.synchronize()
        assert _rel(sums[:, 0].cpu(), ref.sum(0)) < 1e-5


@pytest.mark.parametrize("shape,beta", [((4, 8, 8, 64, 256), 1), ((3, 7, 9, 128, 512), 1), ((2, 8, 8, 64, 256), 0)],
                         ids=["64_256_acc", "128_512_ragged_acc", "64_256_write"])
def test_conv_dgrad_with_bn_tail(cuda, shape, beta):
    """vc_conv_dgrad_bn_tail (the data gradient that completes the gradient of a residual block's output): the
    stored gradient equals vc_conv_dgrad followed by the recorded ReLU decisions (bit-exact), the slab sums equal
    sums over the stored values, and vc_bn_bwd_sums_from_partial reproduces vc_bn_bwd_reduce_mask's reductions."""
    from video_classification_amd import _lib as L
    N, H, W, Cmid, Cout = shape  # forward layer: 1x1 convolution Cout -> Cmid (the next block's conv1); dx has Cout channels
    g = torch.Generator().manual_seed(11)
    M = N * H * W
    dy = torch.randn(M, Cmid, generator=g).to(cuda, torch.bfloat16)
    wt = (torch.randn(Cout, 1, Cmid, generator=g) * 0.1).to(cuda, torch.bfloat16)  # data-gradient pack [ci][taps][co]
    old = torch.randn(M, Cout, generator=g).to(cuda, torch.bfloat16)
    y = (torch.randn(M, Cout, generator=g) * 2 + 0.3).to(cuda, torch.bfloat16)
    bits = torch.randint(0, 256, (M, Cout // 8), generator=g, dtype=torch.uint8).to(cuda)
    desc = L.int_array([N, 1, H, W, Cout, 1, H, W, Cmid, 1, 1, 1, 1, 1, 1, 0, 0, 0])
    assert L.query("vc_conv_dgrad_bn_tail_supported", desc, 1) == 1
    slabs = (M + 63) // 64
    dx0, dx1 = old.clone(), old.clone()
    part = torch.full((slabs, 2, Cout), float("nan"), device=cuda)
    L.call("vc_conv_dgrad", desc, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx0), 0, 0, beta, L.stream())
    L.call("vc_conv_dgrad_bn_tail", desc, 1, L.ptr(dy), L.ptr(wt), L.ptr(dx1), beta, L.ptr(y), L.ptr(bits), L.ptr(part),
           L.stream())
    torch.cuda.synchronize()
    mask = ((bits.cpu().unsqueeze(-1) >> torch.arange(8, dtype=torch.uint8)) & 1).bool().view(M, Cout)
    dz_ref = torch.where(mask, dx0.float().cpu(), torch.zeros(()))
    assert torch.equal(dx1.float().cpu(), dz_ref)
    pad = slabs * 64 - M
    zp = torch.cat([dz_ref, dz_ref.new_zeros(pad, Cout)]).view(slabs, 64, Cout)
    yp = torch.cat([y.float().cpu(), dz_ref.new_zeros(pad, Cout)]).view(slabs, 64, Cout)
    ref = torch.stack([zp.sum(1), (zp * yp).sum(1)], 1)
    assert not torch.isnan(part).any()
    assert _rel(part.cpu(), ref) < 1e-5
    if M % 64 == 0:
        # one group: the reductions of the BatchNorm backward, from the partials and from the separate pass
        mean = y.float().mean(0, keepdim=True).contiguous()
        invstd = (1.0 / (y.float().var(0, unbiased=False, keepdim=True) + 1e-5).sqrt()).contiguous()
        sa = torch.zeros(2, 1, Cout, device=cuda)
        sb = torch.zeros(2, 1, Cout, device=cuda)
        L.call("vc_bn_bwd_sums_from_partial", L.ptr(part), L.ptr(mean), L.ptr(invstd), L.ptr(sa[0]), L.ptr(sa[1]), 1,
               slabs, Cout, L.stream())
        L.call("vc_bn_bwd_reduce_mask", 1, L.ptr(y), L.ptr(dx0), L.ptr(bits), L.ptr(mean), L.ptr(invstd), L.ptr(sb[0]),
               L.ptr(sb[1]), 1, M, Cout, L.stream())
        torch.cuda.synchronize()
        assert torch.equal(dx0, dx1)  # the separate pass masked its copy in place
        assert _rel(sa[0].cpu(), sb[0].cpu()) < 1e-5
        assert _rel(sa[1].cpu(), sb[1].cpu()) < 1e-4


def test_epilogue_side_tensor_forms(cuda):
    """vc_conv_fwd_residual == vc_conv_fwd followed by an add of the residual (one rounding of the sum instead of
    two: within one bf16 ulp of the unfused pair), vc_conv_dgrad_gelu == vc_conv_dgrad times gelu'(pre) computed
    in fp32 from the unfused result."""
    from video_classification_amd import _lib as L
    g = torch.Generator().manual_seed(21)
    M, K, N = 777, 192, 384   # ragged row count, two k-tiles, three column tiles
    x = torch.randn(M, K, generator=g).to(cuda, torch.bfloat16)
    wf = (torch.randn(N, 1, K, generator=g) * 0.1).to(cuda, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(cuda, torch.bfloat16)
    desc = L.int_array([M, 1, 1, 1, K, 1, 1, 1, N, 1, 1, 1, 1, 1, 1, 0, 0, 0])
    assert L.query("vc_conv_epilogue_side_supported", desc, 1, 0) == 1
    y0 = torch.empty(M, N, device=cuda, dtype=torch.float32)
    y1 = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    L.call("vc_conv_fwd", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(bias), L.ptr(y0), 0, 0, 1, 0, L.stream())
    L.call("vc_conv_fwd_residual", desc, 1, L.ptr(x), L.ptr(wf), L.ptr(bias), L.ptr(res), L.ptr(None), 0, L.ptr(y1),
           L.stream())
    torch.cuda.synchronize()
    ref = y0.bfloat16().float() + res.float()  # the kernel parks the bf16-rounded result, then adds in fp32
    assert torch.equal(y1, ref.bfloat16())
    # data gradient of a layer K -> N: dy [M, N] -> dx [M, K], K = 192 > 64 columns
    dy = torch.randn(M, N, generator=g).to(cuda, torch.bfloat16)
    wt = (torch.randn(K, 1, N, generator=g) * 0.1).to(cuda, torch.bfloat16)
    pre = torch.randn(M, K, generator=g).to(cuda, torch.bfloat16)
    assert L.query("vc_conv_epilogue_side_supported", desc, 1, 1) == 1
    d0 = torch.empty(M, K, device=cuda, dtype=torch.bfloat16)
    d1 = torch.empty(M, K, device=cuda, dtype=torch.bfloat16)
    L.call("vc_conv_dgrad", desc, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(d0), 0, 0, 0, L.stream())
    L.call("vc_conv_dgrad_gelu", desc, 1, L.ptr(dy), L.ptr(wt), L.ptr(pre), L.ptr(d1), L.stream())
    torch.cuda.synchronize()
    p = pre.float().cpu().double()
    gp = 0.5 * (1 + torch.erf(p / 2 ** 0.5)) + p * torch.exp(-0.5 * p * p) / (2 * torch.pi) ** 0.5
    ref = d0.float().cpu().double() * gp
    assert _rel(d1.float().cpu(), ref.float()) < 1e-2
    # forward with two outputs: y identical to the plain forward, and gelu of y as stored (bf16)
    yb = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    y2 = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    gl = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    L.call("vc_conv_fwd", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(bias), L.ptr(yb), 0, 0, 0, 0, L.stream())
    L.call("vc_conv_fwd_gelu_dual", desc, 1, L.ptr(x), L.ptr(wf), L.ptr(bias), L.ptr(y2), L.ptr(gl), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(y2, yb)
    h = yb.float().cpu().double()
    want = (0.5 * h * (1 + torch.erf(h / 2 ** 0.5))).float()
    # erff on the device against erf in double: a few fp32 ulps before the rounding to bf16
    assert _rel(gl.float().cpu(), want) < 4e-3
    assert (gl.float().cpu() - want.bfloat16().float()).abs().max() <= 2 * want.abs().max() * 2 ** -8


BENCH_GEOMETRIES = [
    # the layer geometries of the bench workload (512 frames of 224 x 224 through ResNet-50; 32 Conv3D clips)
    # name, N, D, H, W, Ci, Co, k, s, p
    ("l1_1x1_64_256", 512, 1, 56, 56, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("l1_3x3_64_64", 512, 1, 56, 56, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("l2_3x3_s2_128_128", 512, 1, 56, 56, 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("l2_ds_1x1_s2_256_512", 512, 1, 56, 56, 256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("l4_3x3_512_512", 512, 1, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("stem_unfolded_7x1_24_64", 512, 1, 224, 112, 24, 64, (1, 7, 1), (1, 2, 1), (0, 3, 0)),
    ("conv3d_layer2_32_64", 32, 16, 56, 56, 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # fewer source channels than a k-tile is deep (8 / 16 / 32) under several filter taps: the width-unfolded first
    # layers of Conv3D and EncoderCNN, EncoderCNN's second layer, and ragged small ones (odd tap counts, channel and
    # row tails)
    ("conv3d_layer1_unfolded_8_32", 32, 16, 112, 112, 8, 32, (3, 3, 1), (1, 1, 1), (1, 1, 0)),
    ("cnn_layer1_unfolded_16_32", 512, 1, 224, 110, 16, 32, (1, 5, 1), (1, 2, 1), (0, 0, 0)),
    ("cnn_layer2_32_64", 512, 1, 110, 110, 32, 64, (1, 3, 3), (1, 2, 2), (0, 0, 0)),
    ("taps18_ragged_16_40", 3, 2, 13, 11, 16, 40, (2, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("taps27_ragged_8_136", 2, 5, 9, 7, 8, 136, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("taps25_s2_32_72", 3, 1, 21, 17, 32, 72, (1, 5, 5), (1, 2, 2), (0, 2, 2)),
]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", BENCH_GEOMETRIES, ids=[c[0] for c in BENCH_GEOMETRIES])
def test_mfma_kernels_at_bench_geometries(cuda, case):
    """The MFMA kernels at the sizes the benchmark runs them at (1.6 M pixel rows: every workgroup of the persistent
    kernel walks dozens of tiles in XCD-remapped order, the weight gradient is split over ~1000 workgroups, strided data
    gradients run as one GEMM per output phase) against the plain-FMA kernels on the same bf16 data, on the device.
    fp32 accumulation on both sides, different summation order: 2e-4 of the largest magnitude."""
    from video_classification_amd import _lib as L
    name, N, D, H, W, Ci, Co, k, s, p = case
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()) % 1000)
    desc, (Do, Ho, Wo) = _desc(N, D, H, W, Ci, Co, k, s, p)
    taps = k[0] * k[1] * k[2]
    d = L.int_array(desc)
    dt = torch.bfloat16
    x = torch.randn(N, D, H, W, Ci, device=cuda, generator=g).to(dt)
    dy = torch.randn(N, Do, Ho, Wo, Co, device=cuda, generator=g).to(dt)
    wf = (torch.randn(Co, taps, Ci, device=cuda, generator=g) * 0.1).to(dt)
    wt = wf.permute(2, 1, 0).contiguous()  # data-gradient pack [Ci][taps][Co]
    tol = 2e-4
    # forward (fp32 output), and the bf16 output with the BatchNorm slab sums where the geometry takes that path
    y = torch.empty(N, Do, Ho, Wo, Co, device=cuda, dtype=torch.float32)
    yn = torch.empty_like(y)
    L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y), 0, 0, 1, 0, L.stream())
    L.call("vc_conv_fwd_naive_bf16", d, 0, L.ptr(x), L.ptr(wf), L.ptr(None), L.ptr(yn), 0, 1, L.stream())
    torch.cuda.synchronize()
    assert _rel(y, yn) < tol
    M = N * Do * Ho * Wo
    if M % 64 == 0 and L.query("vc_conv_fwd_stats_supported", d, 1, 0, 0) == 1:
        yb = torch.empty(M, Co, device=cuda, dtype=dt)
        part = torch.full((M // 64, 2, Co), float("nan"), device=cuda)
        L.call("vc_conv_fwd_stats", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(yb), 0, 0, L.ptr(part), L.stream())
        torch.cuda.synchronize()
        assert torch.equal(yb, yn.view(M, Co).to(dt)) or _rel(yb.float(), yn.view(M, Co)) < 8e-3  # one bf16 rounding
        ys = yb.float().view(M // 64, 64, Co)
        assert _rel(part[:, 0], ys.sum(1)) < 1e-5 and _rel(part[:, 1], (ys * ys).sum(1)) < 1e-5
        del yb, part, ys
    del y, yn
    if "stem" not in name and "unfolded" not in name:  # a network's first layer has no data gradient
        dx = torch.empty(N, D, H, W, Ci, device=cuda, dtype=torch.float32)
        dxn = torch.full_like(dx, float("nan"))
        L.call("vc_conv_dgrad", d, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx), 0, 1, 0, L.stream())
        L.call("vc_conv_fwd_naive_bf16", d, 1, L.ptr(dy), L.ptr(wt), L.ptr(None), L.ptr(dxn), 0, 1, L.stream())
        torch.cuda.synchronize()
        assert _rel(dx, dxn) < tol
        del dx, dxn
    dw = torch.zeros(Co, taps, Ci, device=cuda, dtype=torch.float32)
    dwn = torch.zeros_like(dw)
    db = torch.zeros(Co, device=cuda, dtype=torch.float32)
    L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dw), L.ptr(db), 0, L.stream())
    L.call("vc_conv_wgrad", d, 1, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dwn), L.ptr(None), 1, L.stream())
    torch.cuda.synchronize()
    assert _rel(dw, dwn) < tol
    assert _rel(db, dy.float().sum(dim=(0, 1, 2, 3))) < tol

@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 16, 16, 64, 64, 1, 2), (2, 8, 32, 256, 128, 1, 1), (4, 16, 16, 64, 64, 3, 2),
                                   (2, 16, 16, 128, 192, 3, 1)])
def test_conv_dgrad_with_bn_pre(cuda, shape):
    """vc_conv_dgrad_bn_pre (the data gradient whose result is the gradient of relu(bn(y))): the stored gradient equals
    vc_conv_dgrad masked by y * scale + shift > 0 (bit-exact), the slab sums equal sums over the stored values, and
    vc_bn_bwd_sums_from_partial reproduces vc_bn_bwd_reduce's reductions per group."""
    from video_classification_amd import _lib as L
    N, H, W, Cin, Cmid, k, G = shape  # forward layer: k x k convolution Cin -> Cmid, stride 1; dx has Cin channels
    g = torch.Generator().manual_seed(12)
    M = N * H * W
    rpg = M // G
    taps = k * k
    dy = torch.randn(M, Cmid, generator=g).to(cuda, torch.bfloat16)
    wt = (torch.randn(Cin, taps, Cmid, generator=g) * 0.1).to(cuda, torch.bfloat16)  # data-gradient pack [ci][taps][co]
    y = (torch.randn(M, Cin, generator=g) * 2 + 0.3).to(cuda, torch.bfloat16)
    stat = torch.randn(4, G, Cin, generator=g).to(cuda)
    stat[1] = stat[1].abs() + 0.5
    desc = L.int_array([N, 1, H, W, Cin, 1, H, W, Cmid, 1, k, k, 1, 1, 1, 0, k // 2, k // 2])
    assert L.query("vc_conv_dgrad_bn_pre_supported", desc, 1, rpg) == 1
    assert L.query("vc_conv_dgrad_bn_pre_supported", desc, 1, rpg + 8) == 0
    slabs = M // 64
    dx0 = torch.empty(M, Cin, device=cuda, dtype=torch.bfloat16)
    dx1 = torch.empty_like(dx0)
    part = torch.full((slabs, 2, Cin), float("nan"), device=cuda)
    L.call("vc_conv_dgrad", desc, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx0), 0, 0, 0, L.stream())
    L.call("vc_conv_dgrad_bn_pre", desc, 1, L.ptr(dy), L.ptr(wt), L.ptr(dx1), L.ptr(y), L.ptr(stat), rpg, L.ptr(part),
           L.stream())
    torch.cuda.synchronize()
    yf = y.float().view(G, rpg, Cin)
    pos = (torch.addcmul(stat[3].unsqueeze(1), yf, stat[2].unsqueeze(1)) > 0).view(M, Cin)
    # (the kernel uses a fused multiply-add: decisions can differ from the two-rounding form only at exact ties)
    dz_ref = torch.where(pos, dx0.float(), torch.zeros((), device=cuda))
    diff = (dx1.float() != dz_ref)
    assert diff.float().mean().item() < 1e-5, diff.sum().item()
    zp = dx1.float().view(slabs, 64, Cin)
    ref = torch.stack([zp.sum(1), (zp * y.float().view(slabs, 64, Cin)).sum(1)], 1)
    assert not torch.isnan(part).any()
    assert _rel(part.cpu(), ref.cpu()) < 1e-5
    sa = torch.zeros(2, G, Cin, device=cuda)
    sb = torch.zeros(2, G, Cin, device=cuda)
    L.call("vc_bn_bwd_sums_from_partial", L.ptr(part), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(sa[0]), L.ptr(sa[1]), G,
           rpg // 64, Cin, L.stream())
    L.call("vc_bn_bwd_reduce", 1, L.ptr(y), L.ptr(dx0), L.ptr(None), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(stat[2]),
           L.ptr(stat[3]), L.ptr(sb[0]), L.ptr(sb[1]), G, rpg, Cin, 1, L.stream())
    torch.cuda.synchronize()
    assert _rel(sa[0].cpu(), sb[0].cpu()) < 1e-4
    assert _rel(sa[1].cpu(), sb[1].cpu()) < 1e-3  # sum dz*y - mean*sum dz against sum dz*(y - mean): cancellation





@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40, 24, 24, 25, 40), (70, 20, 24, 36, 72), (33, 50, 56, 144, 40)])
@pytest.mark.parametrize("code", [0, 1])
def test_many_tap_relayouts(cuda, shape, code):
    """Layers with many taps (a Linear over a flattened feature map) take LDS-tiled transposing kernels for the forward
    pack, the flat data-gradient pack and the gradient's way back: exact against torch permutes, padding included."""
    from video_classification_amd import _lib as L
    Co, Ci, Cip, taps, Cop = shape
    dt = torch.float32 if code == 0 else torch.bfloat16
    g = torch.Generator().manual_seed(9)
    w = torch.randn(Co, Ci, taps, generator=g).to(cuda)
    wf = torch.full((Co, taps, Cip), float("nan"), device=cuda, dtype=dt)
    L.call("vc_pack_weight", code, L.ptr(w), L.ptr(wf), L.ptr(None), Co, Ci, taps, Cip, Cop, L.stream())
    ref = torch.zeros(Co, taps, Cip, device=cuda)
    ref[:, :, :Ci] = w.permute(0, 2, 1)
    assert torch.equal(wf, ref.to(dt))
    wt = torch.full((taps * Cip, Cop), float("nan"), device=cuda, dtype=dt)
    L.call("vc_pack_weight_flat_t", code, L.ptr(w), L.ptr(wt), Co, Ci, taps, Cip, Cop, L.stream())
    reft = torch.zeros(taps, Cip, Cop, device=cuda)
    reft[:, :Ci, :Co] = w.permute(2, 1, 0)
    assert torch.equal(wt.view(taps, Cip, Cop), reft.to(dt))
    if code == 0:
        dwp = torch.randn(Co, taps, Cip, generator=g).to(cuda)
        keep = dwp.clone()
        dw = torch.randn(Co, Ci, taps, generator=g).to(cuda)
        base = dw.clone()
        L.call("vc_unpack_wgrad", L.ptr(dwp), L.ptr(dw), Co, Ci, taps, Cip, 3, L.stream())
        torch.cuda.synchronize()
        assert torch.equal(dw, base + keep[:, :, :Ci].permute(0, 2, 1))
        assert torch.count_nonzero(dwp[:, :, :Ci]) == 0      # cleared while read
        L.call("vc_unpack_wgrad", L.ptr(keep), L.ptr(dw), Co, Ci, taps, Cip, 0, L.stream())
        torch.cuda.synchronize()
        assert torch.equal(dw, keep[:, :, :Ci].permute(0, 2, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(300, 2048, 192, 2), (512, 8192, 256, 8), (130, 36864, 1024, 32), (512, 36864, 1024, 16)])
def test_forward_with_split_reduction(cuda, shape):
    """vc_conv_fwd_splitk (few output tiles, long reduction: (tile, slice) work units + a pass that adds the slices)
    against vc_conv_fwd: fp32 results agree to summation order, bf16 results with bias and ReLU to one rounding."""
    from video_classification_amd import _lib as L
    M, K, N, S = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(M, K, generator=g).to(cuda, torch.bfloat16)
    wf = (torch.randn(N, 1, K, generator=g) * 0.02).to(cuda, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(cuda)
    desc = L.int_array([M, 1, 1, 1, K, 1, 1, 1, N, 1, 1, 1, 1, 1, 1, 0, 0, 0])
    assert L.query("vc_conv_fwd_splitk_plan", desc, 1) == S
    assert L.query("vc_conv_fwd_splitk_plan", L.int_array([100000, 1, 1, 1, K, 1, 1, 1, N, 1, 1, 1, 1, 1, 1, 0, 0, 0]), 1) == 1
    ws = torch.full((S, M, N), float("nan"), device=cuda)
    for relu, f32 in ((0, 1), (1, 0)):
        dt = torch.float32 if f32 else torch.bfloat16
        y0 = torch.empty(M, N, device=cuda, dtype=dt)
        y1 = torch.empty(M, N, device=cuda, dtype=dt)
        L.call("vc_conv_fwd", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(bias), L.ptr(y0), 0, relu, f32, 0, L.stream())
        L.call("vc_conv_fwd_splitk", desc, 1, L.ptr(x), L.ptr(wf), L.ptr(bias), L.ptr(y1), relu, f32, L.ptr(ws),
               L.stream())
        torch.cuda.synchronize()
        assert not torch.isnan(ws).any()
        assert _rel(y1.float().cpu(), y0.float().cpu()) < (1e-5 if f32 else 8e-3)  # bf16: one ulp at the top


@pytest.mark.parametrize("N,H,Ci,Co,k", [(512, 14, 256, 256, 3), (512, 14, 256, 1024, 1), (128, 56, 64, 64, 3),
                                         (128, 56, 64, 256, 1), (128, 28, 512, 128, 1)],
                         ids=["3x3_256", "1x1_256_1024", "3x3_64", "1x1_one_ktile", "1x1_512_128"])
def test_persistent_kernel_repeats_bit_identically(cuda, N, H, Ci, Co, k):
    """Forward and data gradient (bf16 outputs, no atomics anywhere) of whole-chip launches, twelve times each with a
    256 MiB buffer rewritten in between: every repetition equals the first bit for bit, and the first equals the
    naive kernel within accumulation order.  The persistent kernel's waits leave an output tile's stores in flight
    while the next tile is staged; a wait that let a staged k-tile through early would show up here as a difference
    that comes and goes."""
    from video_classification_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    p = k // 2
    desc, (_, Ho, Wo) = _desc(N, 1, H, H, Ci, Co, (1, k, k), (1, 1, 1), (0, p, p))
    d = L.int_array(desc)
    taps = k * k
    x = torch.randn(N, 1, H, H, Ci, generator=g).to(cuda, torch.bfloat16)
    dy = torch.randn(N, 1, Ho, Wo, Co, generator=g).to(cuda, torch.bfloat16)
    w = (torch.randn(Co, Ci, taps, generator=g) * 0.05).to(cuda)
    b = torch.randn(Co, generator=g).to(cuda)
    wf = torch.empty(Co, taps, Ci, device=cuda, dtype=torch.bfloat16)
    wt = torch.empty(Ci, taps, Co, device=cuda, dtype=torch.bfloat16)
    L.call("vc_pack_weight", 1, L.ptr(w), L.ptr(wf), L.ptr(wt), Co, Ci, taps, Ci, Co, L.stream())
    flush = torch.empty(64 << 20, device=cuda)
    first = None
    for rep in range(12):
        y = torch.empty(N, 1, Ho, Wo, Co, device=cuda, dtype=torch.bfloat16)
        dx = torch.empty(N, 1, H, H, Ci, device=cuda, dtype=torch.bfloat16)
        flush.fill_(float(rep))
        L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(b), L.ptr(y), 0, 1, 0, 0, L.stream())
        L.call("vc_conv_dgrad", d, 1, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx), 0, 0, 0, L.stream())
        torch.cuda.synchronize()
        if first is None:
            first = (y, dx)
            yn = torch.empty(N, 1, Ho, Wo, Co, device=cuda, dtype=torch.float32)
            L.call("vc_conv_fwd_naive_bf16", d, 0, L.ptr(x), L.ptr(wf), L.ptr(b), L.ptr(yn), 1, 1, L.stream())
            torch.cuda.synchronize()
            assert _rel(y.float(), yn) < 8e-3
        else:
            assert torch.equal(y, first[0]), f"forward differs in repetition {rep}"
            assert torch.equal(dx, first[1]), f"data gradient differs in repetition {rep}"
