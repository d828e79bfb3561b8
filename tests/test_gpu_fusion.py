"""Image / density fusion CNN on the device (reference models/swintransformer_RNN/image_density_fusion.py): the
attention-gate kernels against the oracle's ChannelAttention / SpatialAttention, the whole module against the golden
fixtures made from the reference's own class."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as G  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 9, 11, 32, 7), (2, 16, 40, 64, 7), (1, 5, 5, 16, 3)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_attention_gate_vs_oracle(cuda, shape, dt):
    from video_classification_amd import fusion
    N, H, W, C, k = shape
    torch.manual_seed(4)
    ca, sa = O.ChannelAttention(C), O.SpatialAttention(k)
    with torch.no_grad():
        for p in list(ca.parameters()) + list(sa.parameters()):
            p.mul_(3.0)   # gates away from 1/2
    x = torch.randn(N, C, H, W)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    go = torch.randn(N, C, H, W)
    xr = x.clone().requires_grad_(True)
    y = xr * ca(xr)
    ref = torch.relu(y * sa(y))
    (ref * go).sum().backward()

    ws = [p.detach().clone().to(cuda).requires_grad_(True) for p in (ca.fc[0].weight, ca.fc[2].weight, sa.conv.weight)]
    xc = x.permute(0, 2, 3, 1).contiguous().view(N, 1, H, W, C).to(cuda, dt).requires_grad_(True)
    out = fusion.attention_gate(xc, *ws)
    out.backward(go.permute(0, 2, 3, 1).contiguous().view(N, 1, H, W, C).to(cuda, dt))
    torch.cuda.synchronize()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    to_nchw = lambda t: t.float().cpu().view(N, H, W, C).permute(0, 3, 1, 2)  # noqa: E731
    assert G.max_rel(to_nchw(out.detach()), ref.detach()) < tol
    assert G.max_rel(to_nchw(xc.grad), xr.grad) < tol
    for w, p in zip(ws, (ca.fc[0].weight, ca.fc[2].weight, sa.conv.weight)):
        assert G.max_rel(w.grad.cpu(), p.grad) < (2e-4 if dt == torch.float32 else 3e-2)


def _fusion_from_golden(cuda, name, dt):
    from video_classification_amd import fusion
    d = G.load(name)
    m = fusion.create_image_density_fusion(3, 1, 32, 64)
    m.load_state_dict(G.group(d, "p"), strict=True)
    m.to(cuda).set_compute_dtype(dt)
    return d, m


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_image_density_fusion_vs_reference_golden(cuda, dt):
    d, m = _fusion_from_golden(cuda, "image_density_fusion", dt)
    f32 = dt == torch.float32
    m.train()
    out = m(torch.from_numpy(d["in/img"]).to(cuda), torch.from_numpy(d["in/den"]).to(cuda))
    assert out.dtype == torch.float32 and tuple(out.shape) == tuple(d["out_train"].shape)
    (out * torch.from_numpy(d["loss_w"]).to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert G.max_rel(out.detach().cpu(), d["out_train"]) < (1e-4 if f32 else 5e-2)
    grefs = G.group(d, "g")
    # every convolution bias sits in front of a BatchNorm: its true gradient is zero (rounding noise on both sides)
    conv_bias = lambda k: k.endswith(".bias") and ("conv" in k.split(".")[-2] or k in ("initial_conv.0.bias", "final_conv.0.bias"))  # noqa: E731
    errs = {k: G.l2_rel(p.grad.cpu(), grefs[k]) for k, p in m.named_parameters() if not conv_bias(k)}
    # bf16: the gate's own weights see the arg-max decisions of both poolings move under rounding, through a hidden
    # layer of two units (fixture: 32 channels / 16): order of magnitude only; exactness is test_attention_gate_vs_oracle
    att = {k: e for k, e in errs.items() if "_attention." in k}
    rest = {k: e for k, e in errs.items() if "_attention." not in k}
    worst = max(rest, key=rest.get)
    assert rest[worst] < (2e-3 if f32 else 0.3), (worst, rest[worst])
    assert max(att.values()) < (2e-3 if f32 else 1.0), att
    bufs = dict(m.named_buffers())
    for k, v in G.group(d, "a").items():
        if "running_" in k:
            assert G.max_rel(bufs[k].cpu(), v) < (1e-4 if f32 else 3e-2), k
    m.eval()
    with torch.no_grad():
        oe = m(torch.from_numpy(d["in/img"]).to(cuda), torch.from_numpy(d["in/den"]).to(cuda))
    assert G.max_rel(oe.cpu(), d["out_eval"]) < (1e-4 if f32 else 5e-2)


@pytest.mark.gpu
def test_image_density_fusion_batch_wrapper_and_resized_density(cuda):
    d, m = _fusion_from_golden(cuda, "image_density_fusion_batch", torch.float32)
    m.train()
    out = m.forward_batch(torch.from_numpy(d["in/x"]).to(cuda), torch.from_numpy(d["in/den"]).to(cuda))
    (out * torch.from_numpy(d["loss_w"]).to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert tuple(out.shape) == tuple(d["out_train"].shape)
    assert G.max_rel(out.detach().cpu(), d["out_train"]) < 1e-4
    assert G.l2_rel(m.multi_scale_aggregation.conv7x7.weight.grad.cpu(), d["g/multi_scale_aggregation.conv7x7.weight"]) < 2e-3


def test_fusion_state_dict_matches_reference_keys():
    """CPU: same keys, shapes and dtypes as the reference's class (read from the fixture it wrote)."""
    from video_classification_amd import fusion
    d = G.load("image_density_fusion")
    ref = G.group(d, "p")
    sd = fusion.create_image_density_fusion(3, 1, 32, 64).state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape) and v.dtype == ref[k].dtype, k


@pytest.mark.gpu
def test_fusion_at_frame_size_and_in_the_loop_body(cuda):
    """224 x 224 frames with the 32 x 32 density map of the reference's configuration
    (UCF101_SwinCRNN_fixed.py:91-98), and the loop body of :232-272: Swin features, fused features and velocity maps
    concatenated for a decoder of the matching width."""
    from video_classification_amd import fusion, modules as M
    torch.manual_seed(1)
    B, T = 1, 3
    fus = fusion.create_image_density_fusion(3, 1, 16, 24).to(cuda)
    X = torch.randn(B, T, 3, 224, 224, device=cuda)
    den = torch.rand(B, T, 1, 32, 32, device=cuda)
    vel = torch.rand(B, T - 1, 1, 4, 4, device=cuda)
    enc = lambda x: x.mean((3, 4)).repeat(1, 1, 4)          # stand-in encoder: [B, T, 12]  # noqa: E731
    dec = M.DecoderRNN(CNN_embed_dim=12 + 24 + 16, h_RNN_layers=1, h_RNN=32, h_FC_dim=16, drop_p=0.0, num_classes=5).to(cuda)
    fus.train()
    dec.train()
    out = fusion.forward_with_proxies(enc, dec, X, fus, den, vel)
    assert tuple(out.shape) == (B, 5)
    out.sum().backward()
    torch.cuda.synchronize()
    g = fus.initial_conv[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    f = fusion.combined_features(torch.zeros(B, T, 12, device=cuda), None, vel)
    assert tuple(f.shape) == (B, T, 28) and torch.equal(f[:, -1, 12:], f[:, -2, 12:])   # last velocity map repeated
