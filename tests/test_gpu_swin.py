"""GPU parity of the Swin path.

* swin_head_tiny: the reference's own SwinTransformerEncoder.forward (per-frame loop, fc/bn/relu head) run with a
  stand-in backbone (golden vector from tests/golden/make_golden.py) -- pins the head and the frame loop.
* Swin-T backbone / ClusterSwin + TCG-LSTM: against the CPU oracle with identical weights.  The backbone
  restates timm's architecture (timm is absent from the reference checkout and this image): PARITY UNPINNED by
  the reference for that sub-network.

Tolerances as in test_gpu_models.py (fp32 2e-4 / 1e-3, bf16 4e-2 / 2e-1).
"""
import pytest
import torch
import torch.nn as nn

import golden_util as G

pytestmark = pytest.mark.gpu
TOL = {torch.float32: (2e-4, 1e-3), torch.bfloat16: (4e-2, 2e-1)}
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]
SMALL = dict(img=56, patch=4, dim=32, depths=(2, 2), heads=(1, 2), ws=7, drop_path_rate=0.0)  # stages at 14x14 (shifted) and 7x7


class _LinearBackbone(nn.Sequential):
    """nn.Sequential(Flatten, Linear(3*8*8, 768)) evaluated by the HIP GEMM on channels-last frames."""
    num_features = 768

    def run(self, a):
        from video_classification_amd import ops
        lin = self[1]
        y = ops.conv(a, lin.weight, lin.bias, (1, 8, 8), 1, 0, ci=3)
        return y.view(y.shape[0], y.shape[-1])


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_swin_encoder_head_and_frame_loop(cuda, dt):
    from video_classification_amd.swin import SwinTransformerEncoder
    d = G.load("swin_head_tiny")
    enc = SwinTransformerEncoder(64, 48, 0.0, 32, swin_kwargs=SMALL)
    enc.swin = _LinearBackbone(nn.Flatten(), nn.Linear(3 * 8 * 8, 768))
    enc._make_head(768, 64, 48, 32)
    enc.load_state_dict(G.group(d, "p"), strict=True)
    enc.to(cuda).set_compute_dtype(dt)
    tol_o, tol_g = TOL[dt]
    x = torch.from_numpy(d["in/x"]).to(cuda)
    enc.train()
    out = enc(x)
    (out * torch.from_numpy(d["loss_w"]).to(cuda)).sum().backward()
    assert G.max_rel(out.detach().cpu(), d["out_train"]) < tol_o
    G.check_grads(dict(enc.named_parameters()), G.group(d, "g"), tol_g, cancelled=["swin.1.bias", "fc1.bias", "fc2.bias"])
    bufs = dict(enc.named_buffers())
    for k, bref in G.group(d, "a").items():
        if k.endswith("num_batches_tracked"):
            assert int(bufs[k]) == int(bref), k
        else:
            assert G.max_rel(bufs[k].cpu(), bref) < tol_o, k
    enc.eval()
    with torch.no_grad():
        assert G.max_rel(enc(x).cpu(), d["out_eval"]) < tol_o


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_swin_backbone_vs_oracle(cuda, dt):
    from oracle import torch_oracle as O
    from video_classification_amd import ops
    from video_classification_amd.swin import SwinT
    torch.manual_seed(1)
    ref = O.SwinT(**SMALL)
    with torch.no_grad():  # non-trivial LayerNorm affine and bias tables
        for n, p in ref.named_parameters():
            if "norm" in n or "relative_position_bias_table" in n or n.endswith(".bias"):
                p.copy_(torch.randn_like(p) * 0.2 + (1.0 if n.endswith("norm.weight") or "norm1.weight" in n or "norm2.weight" in n else 0.0))
    mine = SwinT(**SMALL)
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda)
    x = torch.randn(3, 3, 56, 56, generator=torch.Generator().manual_seed(2))
    w = torch.randn(3, 64, generator=torch.Generator().manual_seed(3))
    out_r = ref(x)
    (out_r * w).sum().backward()
    a = ops.clip_to_frames(x.unsqueeze(0).to(cuda), dt)   # [B=1,T=3] -> 3 frames
    mine.train()
    out_m = mine.run(a)
    (out_m.float() * w.to(cuda)).sum().backward()
    tol_o, tol_g = TOL[dt]
    assert G.max_rel(out_m.detach().float().cpu(), out_r.detach()) < tol_o
    G.check_grads(dict(mine.named_parameters()), {k: p.grad for k, p in ref.named_parameters()}, tol_g)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_cluster_swin_tcg_vs_oracle(cuda, dt):
    """clusterSwin_TCGlstm wiring (train.py:298-323): ClusterSwinEncoder -> (g, c) -> TCG_LSTM."""
    from oracle import torch_oracle as O
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    from video_classification_amd import swin as S
    torch.manual_seed(4)
    ref = O.SwinTCGModel(O.ClusterSwinEncoder(embed_dim=24, k_tokens=8, swin_kwargs=SMALL),
                         O.TCG_LSTM(24, 32, 2, 8, 4, 101, 0.0))
    mine = M.SwinTCGModel(S.ClusterSwinEncoder(embed_dim=24, k_tokens=8, swin_kwargs=SMALL),
                          M.TCG_LSTM(24, 32, 2, 8, 4, 101, 0.0))
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    x = torch.randn(2, 8, 3, 56, 56, generator=torch.Generator().manual_seed(5))
    y = torch.tensor([[7], [93]])
    ref.train()
    out_r = ref(x)
    loss_r = nn.CrossEntropyLoss()(out_r, y.squeeze())
    loss_r.backward()
    mine.train()
    out_m = mine(x.to(cuda))
    loss_m = ops.cross_entropy(out_m, y.to(cuda))
    loss_m.backward()
    tol_o, tol_g = TOL[dt]
    assert G.max_rel(out_m.detach().cpu(), out_r.detach()) < tol_o
    assert abs(loss_m.item() - loss_r.item()) < (1e-4 if dt == torch.float32 else 2e-2)
    G.check_grads(dict(mine.named_parameters()), {k: p.grad for k, p in ref.named_parameters()}, tol_g)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_cluster_swin_encoder_around_standin_backbone(cuda, dt):
    """Golden from the reference's own ClusterSwinEncoder (model_components.py:408-475: per-frame loop, projection,
    k repeated tokens, stacking) run around a stand-in backbone; the HIP module builds the same backbone."""
    from video_classification_amd.swin import ClusterSwinEncoder
    d = G.load("cluster_swin_standin")
    enc = ClusterSwinEncoder(embed_dim=24, k_tokens=8, swin_kwargs=G.SWIN_STANDIN)
    enc.load_state_dict(G.group(d, "p"), strict=True)
    enc.to(cuda).set_compute_dtype(dt)
    enc.train()
    g, c = enc(torch.from_numpy(d["in/x"]).to(cuda))
    assert tuple(g.shape) == (2, 4, 24) and tuple(c.shape) == (2, 4, 8, 24)
    ((g * torch.from_numpy(d["loss_wg"]).to(cuda)).sum() + (c * torch.from_numpy(d["loss_wc"]).to(cuda)).sum()).backward()
    tol_o, tol_g = TOL[dt]
    assert G.max_rel(g.detach().cpu(), d["out_g"]) < tol_o
    assert G.max_rel(c.detach().cpu(), d["out_c"]) < tol_o
    G.check_grads(dict(enc.named_parameters()), G.group(d, "g"), tol_g)


@pytest.mark.gpu
def test_residual_epilogue_with_sample_factors(cuda):
    """vc_conv_fwd_residual with a row scale: y = residual + f[sample] * (x W^T + b), the factors being exactly those
    vc_drop_path applies for the same seed (0 or 1 / (1 - p) per sample)."""
    from video_classification_amd import _lib as L
    import ctypes
    g = torch.Generator().manual_seed(33)
    S_, rows, K, N = 12, 49, 192, 192          # 12 samples of 49 rows: sample boundaries inside the 128-row tiles
    M = S_ * rows
    x = torch.randn(M, K, generator=g).to(cuda, torch.bfloat16)
    wf = (torch.randn(N, 1, K, generator=g) * 0.1).to(cuda, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(cuda, torch.bfloat16)
    desc = L.int_array([M, 1, 1, 1, K, 1, 1, 1, N, 1, 1, 1, 1, 1, 1, 0, 0, 0])
    p, seed = 0.4, ctypes.c_uint64(12345)
    scales = torch.empty(S_, device=cuda)
    L.call("vc_drop_path_scales", L.ptr(scales), S_, p, seed, L.ptr(None), L.stream())
    ones = torch.ones(M, 8, device=cuda, dtype=torch.bfloat16)
    kept = torch.empty_like(ones)
    L.call("vc_drop_path", 1, L.ptr(ones), L.ptr(kept), L.i64(ones.numel()), L.i64(rows * 8), p, seed, L.ptr(None),
           L.stream())
    y0 = torch.empty(M, N, device=cuda, dtype=torch.float32)
    y1 = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    L.call("vc_conv_fwd", desc, 1, L.ptr(x), 0, L.ptr(wf), L.ptr(bias), L.ptr(y0), 0, 0, 1, 0, L.stream())
    L.call("vc_conv_fwd_residual", desc, 1, L.ptr(x), L.ptr(wf), L.ptr(bias), L.ptr(res), L.ptr(scales), rows, L.ptr(y1),
           L.stream())
    torch.cuda.synchronize()
    sc = scales.cpu()
    keepf = torch.tensor(1.0) / torch.tensor(1.0 - p)   # fp32, as the kernel computes it
    assert set(sc.tolist()) <= {0.0, keepf.item()} and 0.0 in sc.tolist() and sc.max() > 0
    assert torch.equal(kept.float()[:, 0].cpu().view(S_, rows)[:, 0], sc.bfloat16().float())  # same decisions
    f = scales.repeat_interleave(rows).unsqueeze(1)
    # the kernel parks the bf16-rounded product and finishes with one fp32 fused multiply-add
    ref = (res.double() + y0.bfloat16().double() * f.double()).float().bfloat16()
    bad = (y1 != ref)
    assert bad.float().mean().item() < 1e-3, bad.sum().item()   # (ties of the double rounding)
    assert G.max_rel(y1.float().cpu(), ref.float().cpu()) < 1e-3


@pytest.mark.gpu
def test_swin_block_stochastic_depth_fused_equals_unfused(cuda):
    """A Swin block in training with drop_path 0.5: the fused form (factors applied in the projection / fc2 GEMM
    epilogues) and the unfused form (scaling pass + add pass) draw the same per-frame decisions from the same seed and
    give the same output and gradients up to the rounding of the intermediate the unfused form stores."""
    from video_classification_amd import ops
    from video_classification_amd.swin import SwinBlock
    torch.manual_seed(3)
    blk = SwinBlock(128, 14, 4, 7, 3).to(cuda)
    F_ = 16
    x = torch.randn(F_, 14, 14, 128, device=cuda).bfloat16()
    go = torch.randn(F_, 14, 14, 128, device=cuda).bfloat16()
    res = {}
    for fused in (True, False):
        ops.FUSE_EPILOGUE_SIDE = fused
        try:
            ops.manual_seed(77)
            blk.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            out = blk.run(xi, 0.5, True)
            out.backward(go.clone())
            torch.cuda.synchronize()
            res[fused] = [out.detach().float().cpu(), xi.grad.float().cpu()] + \
                [p.grad.float().cpu() for p in blk.parameters()]
        finally:
            ops.FUSE_EPILOGUE_SIDE = True
    a, b = res[True][0], res[False][0]
    untouched = ((b - x.float().cpu()).abs().amax(dim=(1, 2, 3)) == 0)   # frames whose two branches were both dropped
    assert 0 < (a - x.float().cpu()).abs().amax(dim=(1, 2, 3)).eq(0).sum() == untouched.sum() < F_
    for u, v in zip(res[True], res[False]):
        assert G.max_rel(u, v) < 2e-2


def test_swin_rnn_deterministic_mode(cuda):
    """ops.set_deterministic on the Swin model: with the relative-position-bias gradient folded in a fixed order (inside a
    block always, across blocks through the workspace) and LayerNorm's parameter gradients summed in order, two steps
    from the same state repeat bit for bit."""
    from video_classification_amd import ops, swin as S
    from video_classification_amd.modules import DecoderRNN, SwinTransformerRNNModel

    def run():
        torch.manual_seed(5)
        enc = S.SwinTransformerEncoder(64, 48, 0.0, 32, swin_kwargs=dict(img=56, patch=4, dim=32, depths=(2, 2),
                                                                          heads=(1, 2), ws=7, drop_path_rate=0.1))
        dec = DecoderRNN(32, 2, 24, 16, 0.0, 9)
        model = SwinTransformerRNNModel(enc, dec).to(cuda)
        model.set_compute_dtype(torch.bfloat16)
        model.train()
        ops.manual_seed(99)
        g = torch.Generator().manual_seed(6)
        x = torch.randn(6, 3, 3, 56, 56, generator=g).to(cuda)
        y = torch.randint(0, 9, (6,), generator=g).to(cuda)
        loss = ops.cross_entropy(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        res = {"loss": loss.detach().float().cpu().reshape(-1)}
        for k, p in model.named_parameters():
            if p.grad is not None:
                res["grad." + k] = p.grad.detach().float().cpu()
        return res

    ops.set_deterministic(True, workspace_mb=64)
    try:
        a, b = run(), run()
    finally:
        ops.set_deterministic(False)
    assert any("relative_position_bias_table" in k for k in a)
    for k in a:
        assert torch.isfinite(a[k]).all(), k
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("ws,H,shift", [(4, 8, 0), (4, 8, 2), (6, 12, 3), (7, 14, 3)])
def test_window_attention_bf16_backward_for_small_windows(cuda, ws, H, shift):
    """Window attention (timm WindowAttention behind model_components.py:412) with windows of 16, 36 and 49 tokens:
    the bf16 MFMA kernels, forward and backward, against the fp32 kernels on the same (bf16-rounded) operands.  The
    backward kernel's counted wait assumes the 24 gradient stores of a full 7 x 7 window; smaller windows issue
    fewer and take the full wait.  Bounds: bf16 rounding of probabilities and outputs, 3e-2 relative L2."""
    from video_classification_amd import ops
    heads, B = 2, 3
    C = heads * 32
    g = torch.Generator().manual_seed(40 + ws)
    qkv = torch.randn(B * H * H, 3 * C, generator=g).to(torch.bfloat16)
    table = (0.5 * torch.randn((2 * ws - 1) ** 2, heads, generator=g))
    dout = torch.randn(B * H * H, C, generator=g).to(torch.bfloat16)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        q = qkv.to(cuda).to(dt).requires_grad_(True)
        t = table.to(cuda).requires_grad_(True)
        out = ops.window_attention(q, t, B, H, H, heads, ws, shift)
        out.backward(dout.to(cuda).to(dt))
        torch.cuda.synchronize()
        res[dt] = (out.detach().float().cpu(), q.grad.float().cpu(), t.grad.float().cpu())
    for a, b, name in zip(res[torch.bfloat16], res[torch.float32], ("out", "dqkv", "dbias")):
        err = ((a - b).norm() / b.norm()).item()
        assert err < 3e-2, (name, err)
