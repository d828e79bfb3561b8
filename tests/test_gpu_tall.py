"""Layers of up to 64 output channels on 256 x 64 tiles (four waves stacked along the pixel rows: VC_NT_TALL unset or 1)
against the 128 x 64 tile (VC_NT_TALL=0; the library reads the variable at every call), through the C ABI.  Same bf16
operands, same MFMA instruction, same k order, one rounding: stored bf16 results bit for bit; the per-slab BatchNorm
sums are folded in another lane order: 1e-5 of the largest magnitude.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


class _Tall:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("VC_NT_TALL")
        os.environ["VC_NT_TALL"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["VC_NT_TALL"]
        else:
            os.environ["VC_NT_TALL"] = self.old


# N, H, W, Ci, Co, k: 3 x 3 with ragged rows / channel tails on one tap / many tiles, whole / 32 output channels
SHAPES = [(5, 13, 11, 64, 64, 3), (3, 9, 7, 136, 40, 1), (16, 28, 28, 256, 64, 1), (7, 12, 12, 128, 32, 3)]


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_forward_and_data_gradient_on_tall_tiles(cuda, shape):
    from video_classification_amd import _lib as L
    N, H, W, Ci, Co, k = shape
    g = torch.Generator().manual_seed(61)
    p, taps, M = k // 2, k * k, N * H * W
    x = torch.randn(M, Ci, generator=g).to(cuda, torch.bfloat16)
    wf = (torch.randn(Co, taps, Ci, generator=g) * 0.1).to(cuda, torch.bfloat16)
    bias = torch.randn(Co, generator=g).to(cuda)
    d = L.int_array([N, 1, H, W, Ci, 1, H, W, Co, 1, k, k, 1, 1, 1, 0, p, p])
    # the data gradient of a layer whose INPUT has Co (<= 64) channels and whose output has Ci: dx has Co channels
    dyt = torch.randn(M, Ci, generator=g).to(cuda, torch.bfloat16)
    wt = (torch.randn(Co, taps, Ci, generator=g) * 0.1).to(cuda, torch.bfloat16)  # [in = Co][taps][out = Ci]
    dt = L.int_array([N, 1, H, W, Co, 1, H, W, Ci, 1, k, k, 1, 1, 1, 0, p, p])
    slabs = (M + 63) // 64
    out = {}
    for tall in (0, 1):
        with _Tall(tall):
            y = torch.full((M, Co), float("nan"), device=cuda, dtype=torch.bfloat16)
            yr = torch.full_like(y, float("nan"))
            ys = torch.full_like(y, float("nan"))
            part = torch.full((slabs, 2, Co), float("nan"), device=cuda)
            dx = torch.full((M, Co), float("nan"), device=cuda, dtype=torch.bfloat16)
            dxa = torch.ones(M, Co, device=cuda, dtype=torch.bfloat16)
            L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(y), 0, 0, 0, 0, L.stream())
            L.call("vc_conv_fwd", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(bias), L.ptr(yr), 0, 1, 0, 0, L.stream())
            if L.query("vc_conv_fwd_stats_supported", d, 1, 0, 0) == 1:
                L.call("vc_conv_fwd_stats", d, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(None), L.ptr(ys), 0, 0, L.ptr(part),
                       L.stream())
            L.call("vc_conv_dgrad", dt, 1, L.ptr(dyt), 0, L.ptr(wt), L.ptr(dx), 0, 0, 0, L.stream())
            L.call("vc_conv_dgrad", dt, 1, L.ptr(dyt), 0, L.ptr(wt), L.ptr(dxa), 0, 0, 1, L.stream())  # accumulate
            torch.cuda.synchronize()
            out[tall] = (y, yr, ys, part, dx, dxa)
    for a, b, what in zip(out[0], out[1], ("y", "relu(y + bias)", "y with sums", "slab sums", "dx", "dx accumulated")):
        if what == "slab sums":
            if not torch.isnan(a).any():
                assert not torch.isnan(b).any()
                assert _rel(b, a) < 1e-5, what
        elif not torch.isnan(a.float()).any():
            assert not torch.isnan(b.float()).any(), what
            assert torch.equal(a, b), what
    yn = torch.empty(M, Co, device=cuda, dtype=torch.float32)
    L.call("vc_conv_fwd_naive_bf16", d, 0, L.ptr(x), L.ptr(wf), L.ptr(None), L.ptr(yn), 0, 1, L.stream())
    torch.cuda.synchronize()
    assert _rel(out[1][0].float(), yn) < 8e-3  # one bf16 rounding of a sum formed in a different order


@pytest.mark.parametrize("shape", [(4, 16, 16, 64, 64, 3, 2), (8, 8, 8, 64, 256, 1, 1), (3, 16, 16, 40, 128, 1, 3)],
                         ids=["3x3_64_two_groups", "1x1_64_from_256", "1x1_40_three_groups"])
def test_data_gradient_with_recomputed_relu_decisions_on_tall_tiles(cuda, shape):
    """vc_conv_dgrad_bn_pre (result of up to 64 channels) on both tile shapes: identical masked gradients, slab sums to
    summation order."""
    from video_classification_amd import _lib as L
    N, H, W, Cin, Cmid, k, G = shape
    g = torch.Generator().manual_seed(62)
    M = N * H * W
    rpg = M // G
    assert M % (64 * G) == 0
    dy = torch.randn(M, Cmid, generator=g).to(cuda, torch.bfloat16)
    wt = (torch.randn(Cin, k * k, Cmid, generator=g) * 0.1).to(cuda, torch.bfloat16)
    y = (torch.randn(M, Cin, generator=g) * 2 + 0.3).to(cuda, torch.bfloat16)
    stat = torch.randn(4, G, Cin, generator=g).to(cuda)
    stat[1] = stat[1].abs() + 0.5
    d = L.int_array([N, 1, H, W, Cin, 1, H, W, Cmid, 1, k, k, 1, 1, 1, 0, k // 2, k // 2])
    assert L.query("vc_conv_dgrad_bn_pre_supported", d, 1, rpg) == 1
    res = {}
    for tall in (0, 1):
        with _Tall(tall):
            dx = torch.full((M, Cin), float("nan"), device=cuda, dtype=torch.bfloat16)
            part = torch.full((M // 64, 2, Cin), float("nan"), device=cuda)
            L.call("vc_conv_dgrad_bn_pre", d, 1, L.ptr(dy), L.ptr(wt), L.ptr(dx), L.ptr(y), L.ptr(stat), rpg, L.ptr(part),
                   L.stream())
            torch.cuda.synchronize()
            res[tall] = (dx, part)
    assert not torch.isnan(res[1][0].float()).any()
    assert torch.equal(res[0][0], res[1][0])
    assert _rel(res[1][1], res[0][1]) < 1e-5
