"""GPU parity of the drop-in modules (HIP kernels through the C ABI) against golden vectors produced by
the reference's own classes (tests/golden/make_golden.py), and against the CPU oracle at the full
configs[0] shape.

Tolerances.  Outputs / buffers: max abs error relative to the largest magnitude of the reference tensor.
Gradients: L2 error relative to the tensor's norm at tol, and max-norm error at 3x tol (golden_util.check_grads).
  float32 compute path : 2e-4 outputs, 1e-3 gradients (different summation order only)
  bfloat16 compute path: 4e-2 outputs, 2e-1 gradients (bf16 storage of activations/weights/activation
                         gradients, fp32 accumulation).  The gradient bound is set by the tensor furthest from the
                         loss: every BatchNorm backward subtracts two projections from the incoming gradient, which
                         amplifies the relative rounding error of what is left, so the error doubles roughly per
                         BatchNorm layer (measured L2 errors on encoder_cnn_tiny: conv4 < conv3 < conv2 < conv1 =
                         0.16); the fp32 path on the same kernels' indexing stays below 1e-3.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

import golden_util as G

pytestmark = pytest.mark.gpu

TOL = {torch.float32: (2e-4, 1e-3), torch.bfloat16: (4e-2, 2e-1)}
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]


def _component(cuda, name, module, input_names, dt, float_inputs_grad=True, cancelled=()):
    d = G.load(name)
    module.load_state_dict(G.group(d, "p"), strict=True)
    module.to(cuda).set_compute_dtype(dt)
    tol_o, tol_g = TOL[dt]
    ins = [torch.from_numpy(d[f"in/{k}"]).to(cuda).requires_grad_(float_inputs_grad) for k in input_names]
    module.train()
    out = module(*ins)
    assert out.dtype == torch.float32
    (out * torch.from_numpy(d["loss_w"]).to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert G.max_rel(out.detach().cpu(), d["out_train"]) < tol_o
    G.check_grads(dict(module.named_parameters()), G.group(d, "g"), tol_g, cancelled)
    if float_inputs_grad:
        for k, x in zip(input_names, ins):
            assert G.max_rel(x.grad.cpu(), d[f"gin/{k}"]) < tol_g, k
    bufs = dict(module.named_buffers())
    for k, bref in G.group(d, "a").items():
        if k.endswith("num_batches_tracked"):
            assert int(bufs[k]) == int(bref), k
        else:
            assert G.max_rel(bufs[k].cpu(), bref) < tol_o, k
    module.eval()
    with torch.no_grad():
        oe = module(*[torch.from_numpy(d[f"in/{k}"]).to(cuda) for k in input_names])
    assert G.max_rel(oe.cpu(), d["out_eval"]) < tol_o


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_encoder_cnn(cuda, dt):
    from video_classification_amd.modules import EncoderCNN
    # the clip itself needs no gradient (and the channel-padded first layer does not produce one)
    _component(cuda, "encoder_cnn_tiny", EncoderCNN(64, 64, 64, 48, 0.0, 32), ["x"], dt, float_inputs_grad=False,
               cancelled=[f"conv{i}.0.bias" for i in (1, 2, 3, 4)])


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_rescnn_encoder_around_standin_trunk(cuda, dt):
    """Golden from the reference's own ResCNNEncoder (constructor wrap, BatchNorm head, per-frame loop:
    model_components.py:164-191) run around a narrow stand-in trunk; the HIP module builds the same trunk."""
    from video_classification_amd import modules as M
    M.RESNET_LAYERS.setdefault("tiny", (1, 1, 1, 1))
    enc = M.ResCNNEncoder(64, 48, 0.0, 32, use_pretrained=False, **G.RESNET_STANDIN)
    if dt == torch.bfloat16:
        # six 64 x 64 frames per BatchNorm group and 2 x 2 maps in the last stage: ill-conditioned with
        # random weights (see test_gpu_fullsize.py on what that does to bf16 gradients); outputs only, then fp32-exact
        d = G.load("rescnn_standin")
        enc.load_state_dict(G.group(d, "p"), strict=True)
        enc.to(cuda).set_compute_dtype(dt).train()
        out = enc(torch.from_numpy(d["in/x"]).to(cuda))
        assert G.max_rel(out.detach().cpu(), d["out_train"]) < 6e-2
        return
    _component(cuda, "rescnn_standin", enc, ["x"], dt, float_inputs_grad=False, cancelled=["fc1.bias", "fc2.bias"])


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_decoder_rnn(cuda, dt):
    from video_classification_amd.modules import DecoderRNN
    _component(cuda, "decoder_rnn_tiny", DecoderRNN(32, 3, 32, 16, 0.0, 101), ["x"], dt)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_decoder_rnn_default_embed_300(cuda, dt):
    """The reference's constructor default CNN_embed_dim = 300 (model_components.py:199) is not a multiple of 8:
    the LSTM input is zero-padded for the 16-byte channel groups of the bf16 GEMMs.  Against the CPU oracle."""
    from oracle import torch_oracle as O
    from video_classification_amd.modules import DecoderRNN
    torch.manual_seed(11)
    ref = O.DecoderRNN(300, 2, 64, 32, 0.0, 101)
    mine = DecoderRNN(300, 2, 64, 32, 0.0, 101)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    x = torch.randn(5, 6, 300, generator=torch.Generator().manual_seed(12))
    w = torch.randn(5, 101, generator=torch.Generator().manual_seed(13))
    xr = x.clone().requires_grad_(True)
    ref.train()
    outr = ref(xr)
    (outr * w).sum().backward()
    xm = x.to(cuda).requires_grad_(True)
    mine.train()
    out = mine(xm)
    (out * w.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    tol_o, tol_g = TOL[dt]
    assert G.max_rel(out.detach().cpu(), outr.detach()) < tol_o
    G.check_grads(dict(mine.named_parameters()), {k: p.grad for k, p in ref.named_parameters()}, tol_g)
    assert tuple(xm.grad.shape) == (5, 6, 300)
    assert G.max_rel(xm.grad.cpu(), xr.grad) < tol_g


def test_lstm_falls_back_beside_side_stream_work(cuda, monkeypatch):
    """While kernels of another stream may share the GPU (overlapped bucket all-reduces), a recurrence whose
    persistent launch would need more than half of the CUs runs on the per-step kernels instead; same results."""
    from video_classification_amd import ops
    torch.manual_seed(2)
    lstm = nn.LSTM(64, 512, 1, batch_first=True, bidirectional=True).to(cuda)
    x = torch.randn(4, 96, 64, device=cuda).to(torch.bfloat16)  # [T, B, In]: 32 x 2 x 3 = 192 workgroups
    cus = torch.cuda.get_device_properties(cuda).multi_processor_count
    assert 192 <= cus < 384, "geometry of this test: 192 workgroups fit the CUs but are more than half of them"
    taken = []
    real = ops.L.call

    def spy(name, *a):
        taken.append(name)
        return real(name, *a)
    monkeypatch.setattr(ops.L, "call", spy)

    def run(xin):
        del taken[:]
        xin = xin.clone().requires_grad_(True)
        for p in lstm.parameters():
            p.grad = None
        y = ops.lstm(xin, lstm)
        y.float().square().sum().backward()
        torch.cuda.synchronize()
        return y.detach().float(), xin.grad.float(), lstm.weight_hh_l0.grad.clone(), list(taken)

    xs = x[:, :64].contiguous()   # 2 batch chunks: 128 workgroups, at most half of the CUs -> stays persistent
    monkeypatch.setattr(ops, "SIDE_STREAM_WORK", False)
    y0, dx0, dw0, calls0 = run(x)
    assert "vc_lstm_persist_fwd" in calls0 and "vc_lstm_persist_bwd" in calls0
    monkeypatch.setattr(ops, "SIDE_STREAM_WORK", True)
    y1, dx1, dw1, calls1 = run(x)
    assert "vc_lstm_persist_fwd" not in calls1 and "vc_lstm_persist_bwd" not in calls1
    assert "vc_lstm_cell_fwd" in calls1 and "vc_lstm_cell_bwd" in calls1
    _, _, _, calls2 = run(xs)
    assert "vc_lstm_persist_fwd" in calls2
    ops.check_lstm_error()
    assert G.max_rel(y1.cpu(), y0.cpu()) < 2e-2
    assert G.l2_rel(dx1.cpu(), dx0.cpu()) < 3e-2
    assert G.l2_rel(dw1.cpu(), dw0.cpu()) < 3e-2


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_tcg_lstm(cuda, dt):
    from video_classification_amd.modules import TCG_LSTM
    _component(cuda, "tcg_lstm_tiny", TCG_LSTM(24, 32, 2, 8, 4, 101, 0.0), ["g", "c"], dt)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_conv3d(cuda, dt):
    from video_classification_amd.modules import Conv3DModel
    _component(cuda, "conv3d_tiny", Conv3DModel(G.conv3d_config()), ["x"], dt, float_inputs_grad=False,
               cancelled=[f"conv3d.{i}.bias" for i in (0, 4, 8, 12)])


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_crnn_train_step_and_eval(cuda, dt):
    """One batch through train_epoch / validate_epoch with the fused CE head and fused Adam(wd=1e-4)."""
    from video_classification_amd import trainer as TR
    from video_classification_amd.modules import CRNNModel
    d = G.load("crnn_tiny_step")
    model = CRNNModel(G.tiny_crnn_config())
    model.load_state_dict(G.group(d, "p"), strict=True)
    model.to(cuda).set_compute_dtype(dt)
    tol_o, tol_g = TOL[dt]
    x, y = torch.from_numpy(d["in/x"]), torch.from_numpy(d["in/y"])
    opt = TR.get_optimizer(model, "adam", lr=1e-3, weight_decay=1e-4)
    crit = TR.FusedCrossEntropyLoss()
    captured = []
    hook = model.register_forward_hook(lambda m, i, o: captured.append(o.detach().float().cpu()))
    loss, acc = TR.train_epoch(model, [(x, y)], crit, opt, cuda, 1)
    assert abs(loss - float(d["train_loss"])) < (1e-4 if dt == torch.float32 else 2e-2)
    assert G.max_rel(captured[0], d["train_logits"]) < tol_o
    if dt == torch.float32:
        assert acc == float(d["train_acc"])
    G.check_grads(dict(model.named_parameters()), G.group(d, "g"), tol_g,
                  cancelled=[f"encoder.conv{i}.0.bias" for i in (1, 2, 3, 4)])
    sd = model.state_dict()
    if dt == torch.float32:
        for k, aref in G.group(d, "a").items():
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(aref), k
            else:
                # Adam's first step moves every weight by ~lr whatever the gradient scale: absolute bound
                assert (sd[k].cpu().double() - aref.double()).abs().max().item() < 2e-4, k
    vloss, vacc = TR.validate_epoch(model, [(x, y)], crit, cuda, 1)
    hook.remove()
    if dt == torch.float32:
        assert abs(vloss - float(d["val_loss"])) < 1e-3
        assert G.max_rel(captured[1], d["val_logits"]) < 1e-3
        assert torch.equal(captured[1].argmax(1), torch.from_numpy(d["val_logits"]).argmax(1))
        assert vacc == float(d["val_acc"])


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_crnn_full_config0(cuda, dt):
    """configs[0] shape: B=4, T=16, 224x224, 1024/768/512 encoder, 3x512 LSTM, 101 classes."""
    from video_classification_amd.modules import CRNNModel
    d = G.load("crnn_cfg0")
    torch.manual_seed(0)
    model = CRNNModel(G.full_crnn_config())
    for k, v in model.state_dict().items():  # same default-init stream as the reference's classes
        assert abs(v.double().abs().sum().item() - float(d[f"psum/{k}"])) <= 1e-9 * max(1.0, float(d[f"psum/{k}"])), k
    model.to(cuda).set_compute_dtype(dt)
    x = torch.randn(4, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1234)).to(cuda)
    y = torch.from_numpy(d["in/y"]).to(cuda)
    tol_o, tol_g = TOL[dt]
    from video_classification_amd import ops
    model.train()
    # fp32: ordered sums (ops.set_deterministic), so that the run repeats bit for bit.  With fp32 atomics the last bit
    # of the BatchNorm statistics changes from run to run, and with 25 M activations in conv1 alone a few sit within
    # that bit of a ReLU's zero: runs landed at 1.0e-5, 1.5e-4 or 2.6e-3 of the reference, nothing in between.
    if dt == torch.float32:
        ops.set_deterministic(True, workspace_mb=256)
    try:
        out = model(x)
        stats = {}
        loss = ops.cross_entropy(out, y, stats)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        if dt == torch.float32:
            ops.set_deterministic(False)
    assert G.max_rel(out.detach().cpu(), d["logits"]) < tol_o
    assert abs(loss.item() - float(d["loss"])) < (1e-4 if dt == torch.float32 else 2e-2)
    if dt == torch.float32:
        assert np.array_equal(stats["pred"].cpu().numpy(), d["argmax"])
        assert np.array_equal(out.argmax(1).cpu().numpy(), d["argmax"])
    else:
        # bf16: the prediction may only differ where the reference's own top-2 margin is inside the bf16 error
        ref = torch.from_numpy(d["logits"])
        top2 = ref.topk(2, dim=1).values
        # output bound 2e-2 at this shape (half the general bf16 bound): all four rows of the fixture have a top-2
        # margin of 4.2-4.4 % of the largest logit, so every prediction is compared
        err_o = G.max_rel(out.detach().cpu(), d["logits"])
        print(f"\n  bf16 logits max-rel {err_o:.3e}")
        assert err_o < 2e-2
        safe = (top2[:, 0] - top2[:, 1]) > 2 * 2e-2 * ref.abs().max()
        print(f"  predictions compared: {int(safe.sum())} of {ref.shape[0]} rows")
        assert int(safe.sum()) >= ref.shape[0] // 2
        assert torch.equal(stats["pred"].cpu()[safe], ref.argmax(1)[safe])
    for k, p in model.named_parameters():
        ref = float(d[f"gnorm/{k}"])
        if k.startswith("encoder.conv") and k.endswith(".0.bias"):
            # bias straight into BatchNorm: true gradient is zero, both sides hold rounding noise
            wref = float(d[f"gnorm/{k[:-4]}weight"])
            assert p.grad.double().norm().item() <= tol_g * wref, k
            continue
        assert abs(p.grad.double().norm().item() - ref) <= (tol_g * ref + 1e-12), k
    assert G.l2_rel(model.decoder.fc2.weight.grad.cpu(), d["g/decoder.fc2.weight"]) < tol_g
    assert G.l2_rel(model.encoder.conv1[0].weight.grad.cpu(), d["g/encoder.conv1.0.weight"]) < tol_g
    model.eval()
    with torch.no_grad():
        oe = model(x)
    assert G.max_rel(oe.cpu(), d["logits_eval"]) < tol_o
    if dt == torch.float32:
        assert torch.equal(oe.argmax(1).cpu(), torch.from_numpy(d["logits_eval"]).argmax(1))


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_resnet_crnn_vs_oracle(cuda, dt):
    """ResNetCRNN (bottleneck trunk [1,1,1,1] for speed of the CPU side, real widths) against the CPU oracle
    with identical weights.  The trunk restates torchvision's architecture: parity unpinned by the
    reference for this sub-network (torchvision is absent from the reference checkout and this image)."""
    import types
    from oracle import torch_oracle as O
    from video_classification_amd import modules as M
    O.RESNET_LAYERS["tiny"] = (1, 1, 1, 1)
    M.RESNET_LAYERS["tiny"] = (1, 1, 1, 1)
    cfg = types.SimpleNamespace(CNN_fc_hidden1=64, CNN_fc_hidden2=48, CNN_embed_dim=32, dropout_p=0.0,
                                use_pretrained=False, RNN_hidden_layers=2, RNN_hidden_nodes=32, RNN_FC_dim=16,
                                num_classes=101, arch="tiny")
    torch.manual_seed(3)
    ref = O.ResNetCRNNModel(cfg)
    mine = M.ResNetCRNNModel(cfg)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    # 6 clips: BatchNorm1d over only 2 samples normalises every feature to +-1 and makes the trunk's
    # gradient vanish analytically, which would turn the comparison into rounding noise vs rounding noise
    x = torch.randn(6, 2, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    y = torch.tensor([[3], [77], [5], [100], [0], [42]])
    ref.train()
    out_r = ref(x)
    loss_r = nn.CrossEntropyLoss()(out_r, y.squeeze())
    loss_r.backward()
    mine.train()
    out_m = mine(x.to(cuda))
    from video_classification_amd import ops
    loss_m = ops.cross_entropy(out_m, y.to(cuda))
    loss_m.backward()
    tol_o, tol_g = TOL[dt]
    assert G.max_rel(out_m.detach().cpu(), out_r.detach()) < tol_o
    assert abs(loss_m.item() - loss_r.item()) < (1e-4 if dt == torch.float32 else 2e-2)
    grefs = {k: p.grad for k, p in ref.named_parameters()}
    # 19 BatchNorm layers whose statistics come from 6 frames amplify rounding (measured 6e-3 on the stem
    # weight in fp32): 3x the per-model bound
    # (fp32: L2 bound 2e-2; single elements of the first layers reach 4.5e-2 of the tensor's maximum)
    G.check_grads(dict(mine.named_parameters()), grefs, 3 * tol_g if dt == torch.bfloat16 else 2e-2,
                  cancelled=["encoder.fc1.bias", "encoder.fc2.bias"])
    br = dict(ref.named_buffers())
    for k, b in mine.named_buffers():
        if not k.endswith("num_batches_tracked"):
            assert G.max_rel(b.cpu(), br[k]) < tol_o, k
        else:
            assert int(b) == int(br[k]), k


def test_resnet_identity_blocks_fused_bn_backward(cuda):
    """Two bottleneck blocks in a row (a projection-shortcut block, then an identity-shortcut block; the pair that
    opens every ResNet stage) in bf16, 2 time steps x 6 frames of 16 x 16 x 64: the data-gradient GEMM of the second
    block's conv1 completes the gradient of the first block's output, applies that block's ReLU decisions and
    leaves the slab sums of its bn3 backward (vc_conv_dgrad_bn_tail).  One forward pass is walked backward twice,
    fusion on and off: gradients equal up to the summation order of the reductions.  Output and gradients are
    also checked against the CPU oracle's blocks run per time step (bf16 bounds of this file's header).

    A pair of blocks rather than a whole trunk: through a whole randomly initialised trunk with batch statistics
    over a few frames the gradient is too ill-conditioned to test -- the oracle's own fp32 and fp64 gradients
    differ by 6e-3 there, rounding the input clip to bf16 moves them by 13-31 %, and two bf16 runs whose
    BatchNorm sums were added in a different order differ by up to 6 % (measured on a [2,2,1,1] trunk, 64 x 64)."""
    from oracle import torch_oracle as O
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    torch.manual_seed(9)
    ds = nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256))
    ref = nn.Sequential(O.Bottleneck(64, 64, 1, ds), O.Bottleneck(256, 64))
    for m in ref.modules():
        if isinstance(m, nn.BatchNorm2d):  # non-trivial affine parameters
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.3, 0.3)
    mine = nn.Sequential(M.Bottleneck(64, 64, 1, nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256))),
                         M.Bottleneck(256, 64))
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda).train()
    B, T, S = 6, 2, 16
    g = torch.Generator().manual_seed(10)
    x = torch.randn(T * B, 64, S, S, generator=g).bfloat16().float()           # frames time-major, NCHW
    dy = torch.randn(T * B, 256, S, S, generator=g).bfloat16().float()
    ref.train()
    xr = x.clone().requires_grad_(True)
    out_r = torch.cat([ref(xr[t * B:(t + 1) * B]) for t in range(T)])
    out_r.backward(dy)
    grefs = {k: p.grad for k, p in ref.named_parameters()}
    xm = x.permute(0, 2, 3, 1).contiguous().view(T * B, 1, S, S, 64).to(cuda, torch.bfloat16).requires_grad_(True)
    out_m = mine[1].run(mine[0].run(xm, T), T)
    dym = dy.permute(0, 2, 3, 1).contiguous().view(T * B, 1, S, S, 256).to(cuda, torch.bfloat16)
    grads = {}
    orig = ops.L.call
    try:
        for fused in (True, False):
            ops.FUSE_BN_TAIL = fused
            fused_calls = []
            ops.L.call = lambda name, *a: (fused_calls.append(name) if name == "vc_conv_dgrad_bn_tail" else None,
                                           orig(name, *a))[1]
            for p in mine.parameters():
                p.grad = None
            xm.grad = None
            out_m.backward(dym.clone(), retain_graph=fused)  # the blocks overwrite the gradient they are handed
            torch.cuda.synchronize()
            assert len(fused_calls) == (1 if fused else 0), fused_calls
            grads[fused] = {k: p.grad.detach().cpu().clone() for k, p in mine.named_parameters()}
            grads[fused]["x"] = xm.grad.detach().float().cpu().view(T * B, S, S, 64).permute(0, 3, 1, 2)
    finally:
        ops.L.call = orig
        ops.FUSE_BN_TAIL = True
    tol_o, tol_g = TOL[torch.bfloat16]
    om = out_m.detach().float().cpu().view(T * B, S, S, 256).permute(0, 3, 1, 2)
    assert G.max_rel(om, out_r.detach()) < tol_o
    grefs["x"] = xr.grad
    for k, gref in grefs.items():
        assert G.l2_rel(grads[True][k], gref) < tol_g, (k, G.l2_rel(grads[True][k], gref))
    diffs = {k: G.l2_rel(gv, grads[False][k]) for k, gv in grads[True].items()}
    worst = max(diffs.values())
    assert worst < 1e-2, sorted(diffs.items(), key=lambda kv: -kv[1])[:6]


def test_lstm_interlayer_dropout_training(cuda):
    """TCG_LSTM with drop=0.1 (configs/clusterSwin_TCGlstm.yaml) in training mode: nn.LSTM's inter-layer
    dropout runs on the library's generator -- not parity-testable against torch's mask stream, so this
    checks determinism under a fixed seed, that eval mode is unaffected, and finite gradients."""
    from video_classification_amd import ops
    from video_classification_amd.modules import TCG_LSTM
    torch.manual_seed(0)
    m = TCG_LSTM(24, 32, 2, 8, 4, 101, 0.1).to(cuda).set_compute_dtype(torch.float32)
    g = torch.randn(2, 16, 24, device=cuda)
    c = torch.randn(2, 16, 8, 24, device=cuda)
    m.train()
    ops.manual_seed(7)
    o1 = m(g, c)
    ops.manual_seed(7)
    o2 = m(g, c)
    o3 = m(g, c)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    o3.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.eval()
    with torch.no_grad():
        e1, e2 = m(g, c), m(g, c)
    assert torch.equal(e1, e2)


def test_graphed_train_step_matches_eager(cuda):
    """The hipGraph replay of the step (zero-grad + fwd + CE + bwd + fused Adam) must produce the weights the
    eager step produces; Adam's bias correction and learning rate are read from device memory at replay time."""
    from video_classification_amd import trainer as TR
    from video_classification_amd.graph import GraphedTrainStep
    from video_classification_amd.modules import CRNNModel
    from video_classification_amd import ops
    d = G.load("crnn_tiny_step")
    x = torch.from_numpy(d["in/x"]).to(cuda)
    y = torch.from_numpy(d["in/y"]).to(cuda)
    results = []
    # ordered sums: Adam turns a gradient element of the size of the atomics' run-to-run noise into a full +-lr
    # update, which would make the comparison below a matter of luck
    ops.set_deterministic(True, workspace_mb=64)
    for graphed in (False, True):
        model = CRNNModel(G.tiny_crnn_config())
        model.load_state_dict(G.group(d, "p"), strict=True)
        model.to(cuda).set_compute_dtype(torch.float32)
        opt = TR.get_optimizer(model, "adam", lr=1e-3, weight_decay=1e-4)
        crit = TR.FusedCrossEntropyLoss()
        model.train()
        losses = []
        if graphed:
            gs = GraphedTrainStep(model, opt, crit, x, y, warmup=2)  # 2 real steps
            for k in range(3):
                if k == 2:
                    opt.param_groups[0]["lr"] = 5e-4  # a scheduler step between replays
                losses.append(gs(x, y).item())
        else:
            for k in range(5):
                if k == 4:
                    opt.param_groups[0]["lr"] = 5e-4
                opt.zero_grad()
                loss = crit(model(x), y)
                loss.backward()
                opt.step()
                if k >= 2:
                    losses.append(loss.item())
        assert opt.steps == 5
        results.append((losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    ops.set_deterministic(False)
    (le, se), (lg, sg) = results
    assert max(abs(a - b) for a, b in zip(le, lg)) < 1e-4
    for k in se:
        assert (se[k].double() - sg[k].double()).abs().max().item() < 1e-5, k


def test_inference_callers(cuda, tmp_path):
    """save_predictions / predict_topk / GraphedInference (the reference's training_utils.py:245-281 and
    predict.py:134-142) on the tiny CRNN in fp32: predictions equal the CPU oracle's eval-mode arg-max, the pickle
    holds the reference's DataFrame columns, top-k probabilities match a softmax of the oracle's logits, and the
    replayed graph returns the eager logits."""
    import pickle
    from oracle import torch_oracle as O
    from video_classification_amd import trainer as TR
    from video_classification_amd.graph import GraphedInference
    from video_classification_amd.modules import CRNNModel
    d = G.load("crnn_tiny_step")
    ref = O.CRNNModel(G.tiny_crnn_config())
    ref.load_state_dict(G.group(d, "p"), strict=True)
    ref.eval()
    model = CRNNModel(G.tiny_crnn_config())
    model.load_state_dict(G.group(d, "p"), strict=True)
    model.to(cuda).set_compute_dtype(torch.float32)
    x, y = torch.from_numpy(d["in/x"]), torch.from_numpy(d["in/y"])
    with torch.no_grad():
        logits_r = ref(x)
    loader = [(x[:2], y[:2]), (x[2:], y[2:])]
    path = tmp_path / "out" / "predictions.pkl"
    df = TR.save_predictions(model, loader, cuda, str(path))
    assert list(df.columns) == ["y", "y_pred"]
    assert df["y"].tolist() == y.flatten().tolist()
    assert df["y_pred"].tolist() == logits_r.argmax(1).tolist()
    with open(path, "rb") as f:
        assert pickle.load(f).equals(df)
    p, idx = TR.predict_topk(model, x[:1].to(cuda), k=5)
    pr = torch.softmax(logits_r[:1], dim=1)[0]
    assert idx.tolist() == torch.sort(pr, descending=True, stable=True).indices[:5].tolist()
    assert G.max_rel(p, pr[idx]) < 1e-4
    infer = GraphedInference(model, x.to(cuda))
    out_g = infer(x.to(cuda)).float().cpu()
    assert G.max_rel(out_g, logits_r) < 2e-4
    x2 = x.flip(0).contiguous()
    assert G.max_rel(infer(x2.to(cuda)).float().cpu(), logits_r.flip(0)) < 2e-4
    # per-script checkpoint layouts (encoder / decoder halves)
    m2 = CRNNModel(G.tiny_crnn_config())
    assert TR.load_split_checkpoint({"cnn_encoder_state_dict": {"module." + k: v for k, v in model.encoder.state_dict().items()},
                                     "rnn_decoder_state_dict": model.decoder.state_dict()}, m2.encoder, m2.decoder)
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(m2.state_dict().values(), model.state_dict().values()))
    assert not TR.load_split_checkpoint({"model_state_dict": {}}, m2.encoder, m2.decoder)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nblocks", [1, 2, 4])
def test_resnet_blocks_bf16_against_emulating_oracle(cuda, nblocks):
    """ResNet-50 stage 2 at its real geometry in bf16 (a strided projection block, then identity blocks; 256 -> 512
    channels, 56 x 56 -> 28 x 28, 2 time steps x 4 frames) against the CPU oracle run with bf16 STORAGE EMULATION
    (oracle/torch_oracle.py rounds to bf16 where the device stores bf16, values and gradients alike).

    One block: what is left between the two is summation order and a few rounding flips -- measured 3.5e-3 on the worst
    gradient tensor (against 8.6e-2 to the fp32 oracle): bound 2e-2, tight enough that a dropped or doubled term in a
    fused epilogue (slab sums, recomputed ReLU masks, phase-decomposed strided data gradient, accumulation into the
    shortcut's gradient buffer) fails it.
    Deeper chains: rounding is a discontinuous map, so ANY difference (here the order of the fp32 atomics of the
    BatchNorm sums) grows from block to block until it saturates at the rounding noise itself: two runs of the HIP path
    on identical inputs differ by 3e-3 / 3e-2 / 1e-1 after 1 / 2 / 4 blocks.  No oracle can be closer to a run than the
    run is to its own repetition, so there the bound is that noise floor, measured in the test over four runs: the
    distance from the emulating oracle to the closest run may be at most twice the widest run-to-run distance (worst
    tensor and median), plus 5e-3."""
    from oracle import torch_oracle as O
    from video_classification_amd import modules as M
    torch.manual_seed(31)

    def stage(mod):
        ds = nn.Sequential(nn.Conv2d(256, 512, 1, stride=2, bias=False), nn.BatchNorm2d(512))
        blocks = [mod.Bottleneck(256, 128, 2, ds)] + [mod.Bottleneck(512, 128) for _ in range(3)]
        return nn.Sequential(*blocks[:nblocks])
    ref = stage(O)
    for m in ref.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.3, 0.3)
    for blk in ref:
        nn.init.uniform_(blk.bn3.weight, 0.2, 0.4)  # residual branches damped, as after training
    B, T, S = 4, 2, 56
    g = torch.Generator().manual_seed(32)
    x = torch.randn(T * B, 256, S, S, generator=g).relu().bfloat16().float()   # a post-ReLU activation, frames time-major
    dy = torch.randn(T * B, 512, S // 2, S // 2, generator=g).bfloat16().float()
    ref.train()
    xr = x.clone().requires_grad_(True)
    O.EMULATE_BF16 = True
    try:
        out_r = torch.cat([ref(xr[t * B:(t + 1) * B]) for t in range(T)])
        out_r.backward(dy)
    finally:
        O.EMULATE_BF16 = False
    want = {k: p.grad for k, p in ref.named_parameters()}
    want["x"], want["out"] = xr.grad, out_r.detach()

    def hip_run():
        mine = stage(M)
        mine.load_state_dict(ref.state_dict(), strict=True)
        mine.to(cuda).train()
        xm = x.permute(0, 2, 3, 1).contiguous().view(T * B, 1, S, S, 256).to(cuda, torch.bfloat16).requires_grad_(True)
        a = xm
        for blk in mine:
            a = blk.run(a, T)
        a.backward(dy.permute(0, 2, 3, 1).contiguous().view(T * B, 1, S // 2, S // 2, 512).to(cuda, torch.bfloat16))
        torch.cuda.synchronize()
        got = {k: p.grad.detach().cpu() for k, p in mine.named_parameters()}
        got["x"] = xm.grad.detach().float().cpu().view(T * B, S, S, 256).permute(0, 3, 1, 2)
        got["out"] = a.detach().float().cpu().view(T * B, S // 2, S // 2, 512).permute(0, 3, 1, 2)
        return got
    # four runs: one pair's distance is itself a noisy estimate of the floor (measured 1.5e-2 .. 3e-2 for the same two
    # blocks), so the floor is the widest pair and the distance to the oracle is that of the closest run
    runs = [hip_run() for _ in range(4)]
    err = {k: min(G.l2_rel(r[k], want[k]) for r in runs) for k in want}
    floor = {k: max(G.l2_rel(runs[j][k], runs[i][k]) for i in range(4) for j in range(i + 1, 4)) for k in want}
    G.report_against_noise_floor(f"{nblocks} block(s), bf16 vs emulating oracle", err, floor)
    if nblocks == 1:
        assert max(err.values()) < 2e-2, max(err, key=err.get)
    G.assert_within_noise_floor(err, floor)


@pytest.mark.gpu
def test_bottleneck_chain_reductions_in_dgrad_epilogue(cuda):
    """Inside a bottleneck conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu -> conv3 the data-gradient kernels of conv2 and
    conv3 also do the backward reductions of bn1 and bn2 (ops.BNPre).  Same gradients as with the separate reduction
    pass (ops.FUSE_BN_PRE = False): the two differ in summation order only."""
    from video_classification_amd import modules as M, ops
    torch.manual_seed(5)
    T, B = 2, 16
    trunk = M.make_resnet_trunk("resnet50", width=64)
    blocks = torch.nn.Sequential(*list(trunk[4])[:2]).to(cuda)   # the downsample block and a plain one
    x = torch.randn(T * B, 1, 16, 16, 64, device=cuda).bfloat16()
    go = torch.randn(T * B, 1, 16, 16, 256, device=cuda).bfloat16()
    grads = {}
    for fused in (True, False, None):   # None: the separate pass once more, for its own run-to-run differences
        ops.FUSE_BN_PRE = fused is True
        try:
            blocks.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            a = xi
            for blk in blocks:
                a = blk.run(a, T)
            a.backward(go.clone())  # the last block owns the gradient it is handed and overwrites it
            torch.cuda.synchronize()
            grads[fused] = {n: p.grad.float().cpu() for n, p in blocks.named_parameters()}
            grads[fused]["x"] = xi.grad.float().cpu()
        finally:
            ops.FUSE_BN_PRE = True
    # fp32 atomics order the BatchNorm sums differently from run to run and two blocks of bf16 rounding amplify that:
    # repetitions of ONE variant measured 6e-3 .. 6e-2 apart (worst tensor), 3e-4 .. 1.6e-2 (median); a wrong mask or
    # sum shows as 0.5 and more.  Exactness of the kernel itself: test_conv_dgrad_with_bn_pre.
    err = {n: G.max_rel(grads[True][n], grads[False][n]) for n in grads[True]}
    floor = {n: G.max_rel(grads[None][n], grads[False][n]) for n in grads[True]}
    G.report_against_noise_floor("dgrad-epilogue reductions vs separate pass", err, floor)
    assert max(err.values()) < 0.15, max(err, key=err.get)
    assert sorted(err.values())[len(err) // 2] < 0.05


@pytest.mark.gpu
def test_deterministic_mode_repeats_bit_for_bit(cuda):
    """trainer.set_deterministic (reference: utils/common/training_utils.py:27): two training steps of the ResNet CRNN
    from the same state give bit-identical losses, gradients and running statistics, and the same values as the default
    mode up to summation order.  Without the mode the same two runs differ (fp32 atomics), which is what makes the
    check meaningful."""
    from video_classification_amd import modules as M, ops, trainer

    import types
    M.RESNET_LAYERS["tiny"] = (1, 1, 1, 1)
    cfg = types.SimpleNamespace(CNN_fc_hidden1=64, CNN_fc_hidden2=48, CNN_embed_dim=32, dropout_p=0.0,
                                use_pretrained=False, RNN_hidden_layers=2, RNN_hidden_nodes=32, RNN_FC_dim=16,
                                num_classes=11, arch="tiny")

    def run():
        torch.manual_seed(3)
        model = M.ResNetCRNNModel(cfg).to(cuda)
        model.set_compute_dtype(torch.bfloat16)
        model.train()
        g = torch.Generator().manual_seed(4)
        x = torch.randn(6, 4, 3, 96, 96, generator=g).to(cuda)
        y = torch.randint(0, 11, (6,), generator=g).to(cuda)
        loss = ops.cross_entropy(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        res = {"loss": loss.detach().float().cpu().reshape(-1)}
        for k, p in model.named_parameters():
            res["grad." + k] = p.grad.detach().float().cpu()
        for k, b in model.named_buffers():
            if b.dtype.is_floating_point:
                res["buf." + k] = b.detach().float().cpu()
        return res

    base1, base2 = run(), run()
    differing = [k for k in base1 if not torch.equal(base1[k], base2[k])]
    assert differing, "the default mode repeated exactly: this test no longer shows what the mode is for"
    trainer.set_deterministic(5)
    try:
        assert ops.is_deterministic()
        det1, det2 = run(), run()
    finally:
        ops.set_deterministic(False)
    assert not ops.is_deterministic()
    for k in det1:
        assert torch.equal(det1[k], det2[k]), k
    # same mathematics as the default mode: what differs is summation order (and what rounding makes of it)
    import golden_util as GU
    worst = max(GU.l2_rel(det1[k], base1[k]) for k in det1)
    floor = max(GU.l2_rel(base2[k], base1[k]) for k in det1)
    assert worst < 2 * floor + 5e-3, (worst, floor)


@pytest.mark.gpu
def test_eval_batchnorm_folded_into_convolutions(cuda):
    """Eval mode without autograd (predict / validate callers): conv -> BatchNorm (-> + identity) -> ReLU of the
    bottleneck blocks runs as one convolution with folded weights.  Same result as the separate BatchNorm pass up to
    bf16 rounding of the folded weights; the fold follows the running statistics when training has moved them."""
    from video_classification_amd import modules as M, ops
    torch.manual_seed(9)
    trunk = M.make_resnet_trunk("resnet50", width=64)
    blocks = torch.nn.Sequential(*list(trunk[4])[:2], *list(trunk[5])[:2]).to(cuda)   # with both projection blocks
    for m in blocks.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.3, 0.3)
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 2.0)
    x = torch.randn(8, 1, 28, 28, 64, device=cuda).relu().bfloat16()

    def forward(fold):
        ops.FOLD_EVAL_BN = fold
        try:
            with torch.no_grad():
                a = x
                for blk in blocks:
                    a = blk.run(a, 2)
            return a.float()
        finally:
            ops.FOLD_EVAL_BN = True
    blocks.eval()
    plain, folded = forward(False), forward(True)
    assert G.l2_rel(folded.cpu(), plain.cpu()) < 2e-2
    assert (plain >= 0).all() and (folded >= 0).all()
    # a training step moves the running statistics through the kernels' raw pointers: the cached fold must follow
    blocks.train()
    a = x.clone().requires_grad_(True)
    out = a
    for blk in blocks:
        out = blk.run(out, 2)
    out.float().sum().backward()
    blocks.eval()
    plain2, folded2 = forward(False), forward(True)
    moved, apart = G.l2_rel(plain2.cpu(), plain.cpu()), G.l2_rel(folded2.cpu(), plain2.cpu())
    print(f"[eval fold] folded vs separate pass {G.l2_rel(folded.cpu(), plain.cpu()):.2e}; after a training step: "
          f"outputs moved by {moved:.2e}, folded vs separate pass {apart:.2e}")
    assert apart < 2e-2 and moved > 4 * apart   # a stale fold would be `moved` away from the separate pass


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["crnn", "conv3d", "tcg"])
def test_deterministic_mode_other_models(cuda, which):
    """The other convolutional / recurrent models in deterministic mode (BatchNorm + max-pool backward, 3-D
    convolutions, the flattened fc1, the TCG-LSTM fusion): two steps from the same state repeat bit for bit."""
    from video_classification_amd import modules as M, ops

    def run():
        torch.manual_seed(11)
        g = torch.Generator().manual_seed(12)
        if which == "crnn":
            model, x = M.CRNNModel(G.tiny_crnn_config()), torch.randn(6, 4, 3, 64, 64, generator=g)
        elif which == "conv3d":
            model, x = M.Conv3DModel(G.conv3d_config()), torch.randn(4, 1, 8, 48, 48, generator=g)
        else:
            model, x = M.TCG_LSTM(24, 32, 2, 8, 4, 7, 0.0), None
        model = model.to(cuda)
        model.set_compute_dtype(torch.bfloat16)
        model.train()
        if which == "tcg":
            out = model(torch.randn(3, 16, 24, generator=g).to(cuda), torch.randn(3, 16, 8, 24, generator=g).to(cuda))
            y = torch.randint(0, 7, (3,), generator=g).to(cuda)
        else:
            out = model(x.to(cuda))
            y = torch.randint(0, 101, (x.shape[0],), generator=g).to(cuda)
        loss = ops.cross_entropy(out, y)
        loss.backward()
        torch.cuda.synchronize()
        res = {"loss": loss.detach().float().cpu().reshape(-1), "out": out.detach().float().cpu()}
        for k, p in model.named_parameters():
            res["grad." + k] = p.grad.detach().float().cpu()
        return res

    ops.set_deterministic(True, workspace_mb=64)
    try:
        a, b = run(), run()
    finally:
        ops.set_deterministic(False)
    for k in a:
        assert torch.isfinite(a[k]).all(), k
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
def test_train_epoch_replaying_the_step_as_a_graph(cuda):
    """train_epoch(graph=True): the first batch runs eagerly once and is captured, later batches of that shape replay
    the hipGraph, the short last batch runs eagerly.  Same losses, accuracy and weights as the eager epoch (ordered
    sums, so that Adam does not turn the atomics' noise into +-lr), over two epochs with a scheduler step between."""
    from video_classification_amd import ops, trainer as TR
    from video_classification_amd.modules import CRNNModel
    d = G.load("crnn_tiny_step")
    g = torch.Generator().manual_seed(21)
    B = d["in/x"].shape[0]
    # three classes only: a third of the predictions is right by chance, so a replay that reported the count of some
    # other call (the criterion's statistics are rebound by every eager call) shows up in the accuracy
    batches = [(torch.randn(*d["in/x"].shape, generator=g), torch.randint(0, 3, (B, 1), generator=g)) for _ in range(4)]
    batches.append((batches[0][0][:2].clone(), batches[0][1][:2].clone()))    # a short last batch
    ops.set_deterministic(True, workspace_mb=64)
    try:
        results = []
        for graph in (False, True):
            model = CRNNModel(G.tiny_crnn_config())
            model.load_state_dict(G.group(d, "p"), strict=True)
            with torch.no_grad():  # predictions fall into the three classes the labels use
                model.decoder.fc2.bias[:3] += 5.0
            model.to(cuda).set_compute_dtype(torch.float32)
            opt = TR.get_optimizer(model, "adam", lr=1e-3, weight_decay=1e-4)
            sched = TR.get_scheduler(opt, "step", step_size=1, gamma=0.5)
            crit = TR.FusedCrossEntropyLoss()
            ops.manual_seed(77)
            stats = []
            for epoch in range(2):
                stats.append(TR.train_epoch(model, batches, crit, opt, cuda, epoch, graph=graph))
                stats.append(TR.validate_epoch(model, batches[:1], crit, cuda, epoch))
                sched.step()
            assert opt.steps == 10
            assert (getattr(model, "_vc_train_graph", None) is not None) == graph
            results.append((stats, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    finally:
        ops.set_deterministic(False)
    (se, we), (sg, wg) = results
    for (le, ae), (lg, ag) in zip(se, sg):
        assert abs(le - lg) < 1e-4 and ae == ag
    assert se[2][1] > 0 and sg[2][1] > 0, "second-epoch training accuracy should not be zero with three classes"
    for k in we:
        assert (we[k].double() - wg[k].double()).abs().max().item() < 1e-5, k
