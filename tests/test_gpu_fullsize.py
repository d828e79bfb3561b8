"""GPU parity at the shapes BASELINE.json quotes: clips of 16 x 3 x 224 x 224 (Conv3D: 1 x 16 x 112 x 112), 101
classes, full-width models, against the CPU oracle with identical weights and inputs -- logits, loss, predictions,
every parameter gradient.  The oracle's forward + backward takes about 10 s per model on 8-16 host cores.

  ResNetCRNN   ResCNNEncoder(ResNet-50 trunk, 1024/768/512) + DecoderRNN(2 x 512, FC 256), configs[1], 6 clips
  ResNetCRNN   the same around the ResNet-152 trunk and a 3 x 512 LSTM (what the reference instantiates), 6 clips
  Swin-RNN     SwinTransformerEncoder(Swin-T, 1024/768/512) + DecoderRNN(3 x 512, FC 256), configs[3], 6 clips
  clusterSwin  ClusterSwinEncoder(Swin-T, E = 96, k = 8) + TCG_LSTM(H = 512, L = 2), configs[4], 4 clips
  Conv3D       Conv3DModel at its native 1 x 16 x 112 x 112, configs[2], 4 clips

The ResNet trunk and the Swin-T backbone restate third-party architectures (torchvision / timm are absent from the
reference checkout and this image): these tests hold the HIP path to the oracle's restatement of them.

Tolerances.  Outputs: max error relative to the largest reference logit, 2e-4 in fp32 and 1e-2 in bf16 (measured
7e-7 / 3.7e-3 at worst).  Gradients: L2 error relative to the tensor's norm, bounds stated at each test from the
measured error and the conditioning of the model (BatchNorm over few samples amplifies any rounding: the tests say
how much the oracle's OWN gradients move when only its input is rounded to bf16).  Predictions: equal in fp32; in
bf16 equal wherever the reference's own top-2 margin exceeds twice the output bound -- every test prints how many rows
that is and requires at least half the batch (with the 1e-2 bound: 4 of 6, 6 of 6, 4 of 4 rows on these seeds).  The
fp32 runs use ordered sums (ops.set_deterministic) and leave out, by index, the head channels whose ReLU input the
ORACLE finds within 1e-5 of zero; everything else in the head and the decoder is held to 1e-4."""
import functools
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import golden_util as G

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]


NO_DROP = dict(drop_path_rate=0.0)  # stochastic depth (timm default 0.1) draws random samples: off for parity

def _compare(mine, out_m, loss_m, ref_out, ref_loss, ref_grads, dt, tol_g, cancelled=(), group_tol=None, exclude=None,
             tol_o=None, min_rows=1):
    """`exclude` = {parameter name: indices along dim 0 left out of the comparison} (see _near_zero_relu_channels).
    Predictions: equal in fp32; in bf16 equal on the rows whose top-2 margin in the reference exceeds twice the output
    bound -- the number of rows compared is printed and must be at least `min_rows`."""
    tol_o = tol_o if tol_o is not None else (2e-4 if dt == torch.float32 else 1e-2)
    min_rows = max(min_rows, ref_out.shape[0] // 2)
    out_m = out_m.detach().float().cpu()
    err_o = G.max_rel(out_m, ref_out)
    errs = {}
    for k, p in mine.named_parameters():
        if any(k.endswith(c) for c in cancelled):
            continue
        gm, gr = p.grad.detach().cpu(), ref_grads[k]
        if exclude and k in exclude and len(exclude[k]):
            keep = torch.ones(gm.shape[0], dtype=torch.bool)
            keep[exclude[k]] = False
            gm, gr = gm[keep], gr[keep]
        errs[k] = G.l2_rel(gm, gr)
    worst = max(errs, key=errs.get)
    groups = {}
    for k, e in errs.items():
        gk = ".".join(k.split(".")[:2])
        groups[gk] = max(groups.get(gk, 0.0), e)
    print("\n  worst gradient error per group:", {k: float(f"{v:.2e}") for k, v in groups.items()})
    print(f"\n[{'f32' if dt == torch.float32 else 'bf16'}] logits max-rel {err_o:.3e}, loss {loss_m:.6f} vs {ref_loss:.6f}, "
          f"worst gradient {worst} {errs[worst]:.3e}, median {sorted(errs.values())[len(errs) // 2]:.3e}")
    assert err_o < tol_o
    assert abs(loss_m - ref_loss) < (1e-4 if dt == torch.float32 else 2e-2)
    if dt == torch.float32:
        assert torch.equal(out_m.argmax(1), ref_out.argmax(1))
        print(f"  predictions compared: all {ref_out.shape[0]} rows (fp32: every row must agree)")
    else:
        top2 = ref_out.topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 2 * tol_o * ref_out.abs().max()
        print(f"  predictions compared: {int(safe.sum())} of {ref_out.shape[0]} rows (top-2 margin above twice the output bound); "
              f"all rows agree: {bool(torch.equal(out_m.argmax(1), ref_out.argmax(1)))}")
        assert int(safe.sum()) >= min_rows, "no row has a margin the bf16 bound can resolve: choose other weights"
        assert torch.equal(out_m.argmax(1)[safe], ref_out.argmax(1)[safe])
    assert errs[worst] < tol_g, (worst, errs[worst])
    for prefix, bound in (group_tol or {}).items():  # tighter bounds for the tensors near the loss
        for k, e in errs.items():
            if k.startswith(prefix):
                assert e < bound, (k, e, bound)
    for c in cancelled:  # biases straight into a BatchNorm: the true gradient is zero, both sides hold rounding noise
        for k, p in mine.named_parameters():
            if k.endswith(c):
                wk = k[:-len("bias")] + "weight"
                assert p.grad.abs().max().item() <= tol_g * ref_grads[wk].abs().max().item(), k


def _emulating(O, on):
    """Context: the oracle rounds to bf16 wherever the HIP path stores bf16 (oracle/torch_oracle.py, EMULATE_BF16)."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        old, O.EMULATE_BF16 = O.EMULATE_BF16, on
        try:
            yield
        finally:
            O.EMULATE_BF16 = old
    return cm()


def _vs_emulation(mine, out, emu, rerun, cancelled=(), at_least=0.0):
    """The bf16 HIP path against the oracle run with bf16 storage emulation, bounded by the path's own run-to-run
    noise floor (`rerun()` repeats forward + backward on the same model and inputs and returns the output): see
    tests/test_gpu_models.py::test_resnet_blocks_bf16_against_emulating_oracle for why no tighter bound exists for a
    deep bf16 network, and for the tight single-block bound.  The Swin models repeat bit for bit up to their fp32 weight
    gradient atomics (LayerNorm has no cross-row sums), so their floor is ~1e-6 and `at_least` states the bound: what
    separates the device from the emulation there is the emulation's own infidelity (rounding inside the attention
    kernel, sums rounded in two steps), amplified by whatever follows it."""
    emu_out, emu_grads = emu
    keep = lambda k: not any(k.endswith(c) for c in cancelled)  # noqa: E731
    g1 = {k: p.grad.detach().cpu().clone() for k, p in mine.named_parameters() if keep(k)}
    o1 = out.detach().float().cpu()
    for p in mine.parameters():
        p.grad = None
    # two repeats, the larger distance counts: one repeat is one draw of the spread (the Conv3D stack gave 0.021, 0.040,
    # 0.043, 0.044 and 0.064 for its first layer in five runs of this test)
    err = {k: G.l2_rel(g1[k], emu_grads[k]) for k in g1}
    err["logits"] = G.l2_rel(o1, emu_out)
    floor = {k: 0.0 for k in err}
    for _ in range(2):
        for p in mine.parameters():
            p.grad = None
        o2 = rerun().detach().float().cpu()
        for k, p in mine.named_parameters():
            if keep(k):
                floor[k] = max(floor[k], G.l2_rel(p.grad.detach().cpu(), g1[k]))
        floor["logits"] = max(floor["logits"], G.l2_rel(o2, o1))
    G.report_against_noise_floor("bf16 vs bf16-emulating oracle", err, floor)
    G.assert_within_noise_floor(err, floor, at_least=at_least)


def _ordered_sums(on):
    """Context: the device path adds in a fixed order (ops.set_deterministic), so that an fp32 comparison is not at the
    mercy of which way the fp32 atomics' rounding pushes a ReLU input that sits at zero."""
    import contextlib
    from video_classification_amd import ops

    @contextlib.contextmanager
    def cm():
        if on:
            ops.set_deterministic(True, workspace_mb=256)
        try:
            yield
        finally:
            if on:
                ops.set_deterministic(False)
    return cm()


@functools.lru_cache(maxsize=None)
def _resnet_ref(emulate=False, arch="resnet50", layers=2, B=6):
    from oracle import torch_oracle as O
    cfg = types.SimpleNamespace(CNN_fc_hidden1=1024, CNN_fc_hidden2=768, CNN_embed_dim=512, dropout_p=0.0,
                                use_pretrained=False, RNN_hidden_layers=layers, RNN_hidden_nodes=512, RNN_FC_dim=256,
                                num_classes=101, arch=arch)
    torch.manual_seed(11)
    ref = O.ResNetCRNNModel(cfg)
    with torch.no_grad():  # residual branches damped, as in a trained network (see the test's docstring)
        for n, p in ref.named_parameters():
            if n.endswith("bn3.weight"):
                p.fill_(0.1)
    g = torch.Generator().manual_seed(1234)
    # image-like clips: a smooth random picture per frame (8 x 8 noise upsampled), a colour cast, fine noise
    low = F.interpolate(torch.randn(B * 16, 3, 8, 8, generator=g), size=(224, 224), mode="bilinear", align_corners=False)
    x = (1.5 * low + torch.randn(B * 16, 3, 1, 1, generator=g) +
         0.3 * torch.randn(B * 16, 3, 224, 224, generator=g)).view(B, 16, 3, 224, 224)
    y = torch.randint(0, 101, (B, 1), generator=g)
    ref.train()
    # inputs of the head's two ReLUs (the BatchNorm1d outputs, unit scale), one call per time step: a value within 1e-5
    # of zero may take the other branch on the device, whose trunk output differs by ~1e-6 (summation order)
    near = {"bn1": set(), "bn2": set()}
    hooks = [getattr(ref.encoder, n).register_forward_hook(
        lambda m, i, o, n=n: near[n].update((o.detach().abs() < 1e-5).any(0).nonzero().flatten().tolist()))
        for n in near]
    with _emulating(O, emulate):
        out = ref(x)
        loss = nn.CrossEntropyLoss()(out, y.squeeze(1))
        loss.backward()
    for h in hooks:
        h.remove()
    _NEAR_ZERO[(emulate, arch, layers, B)] = {k: sorted(v) for k, v in near.items()}
    return cfg, ref.state_dict(), x, y, out.detach(), loss.item(), {k: p.grad for k, p in ref.named_parameters()}


_NEAR_ZERO = {}


def _near_zero_relu_channels(emulate=False, arch="resnet50", layers=2, B=6):
    """{parameter: channel indices} of the head channels whose ReLU input lies within 1e-5 of zero for some sample in
    the ORACLE's forward pass: their ReLU decision, and with it that channel's row of the fc / bn gradients, is not
    determined at the precision two fp32 implementations agree to.  Exactly those channels are left out (and printed)."""
    _resnet_ref(emulate, arch, layers, B)
    nz = _NEAR_ZERO[(emulate, arch, layers, B)]
    print(f"\n  head ReLU inputs within 1e-5 of zero in the oracle: fc1/bn1 channels {nz['bn1']}, fc2/bn2 channels {nz['bn2']}")
    ex = {}
    for bn, fc in (("bn1", "fc1"), ("bn2", "fc2")):
        idx = torch.tensor(nz[bn], dtype=torch.long)
        for k in (f"encoder.{fc}.weight", f"encoder.{bn}.weight", f"encoder.{bn}.bias"):
            ex[k] = idx
    return ex


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_resnet50_crnn_ucf_shape(cuda, dt):
    """Conditioning decides what can be asked here.  With torchvision's default initialisation (every bn3.weight = 1)
    and white-noise clips the oracle itself is chaotic: rounding only the input clip to bf16 moves ITS logits by 3 %
    and its gradients by 12-136 % (50 un-damped residual blocks, then BatchNorm1d over 6 nearly identical pooled
    features per time step).  The test therefore uses image-like clips and bn3.weight = 0.1 (residual branches
    damped, as training leaves them); the oracle's sensitivity to bf16 input rounding is then 6e-4 on the logits,
    2e-3 (decoder) to 5e-2 (LSTM, fc3) to 0.1-0.2 (trunk) on the gradients.  Gradient bounds: fp32 2e-2, bf16 0.5."""
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    cfg, sd, x, y, ref_out, ref_loss, ref_grads = _resnet_ref()
    f32 = dt == torch.float32
    mine = M.ResNetCRNNModel(cfg)
    mine.load_state_dict(sd, strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    mine.train()
    with _ordered_sums(f32):
        out = mine(x.to(cuda))
        loss = ops.cross_entropy(out, y.to(cuda))
        loss.backward()
        torch.cuda.synchronize()
    ops.check_lstm_error()
    # measured: fp32 trunk 7e-3, head and decoder < 1e-5; bf16 decoder.fc2 9e-3, decoder 5e-2, head 0.25, trunk 0.40
    _compare(mine, out, loss.item(), ref_out, ref_loss, ref_grads, dt, 2e-2 if f32 else 0.7,
             cancelled=["encoder.fc1.bias", "encoder.fc2.bias"],
             group_tol={"decoder.": 1e-4, "encoder.fc": 1e-4, "encoder.bn": 1e-4} if f32 else
             {"decoder.fc2": 3e-2, "decoder.": 0.12},
             exclude=_near_zero_relu_channels() if f32 else None)
    if not f32:
        e = _resnet_ref(True)

        def rerun():
            o = mine(x.to(cuda))
            ops.cross_entropy(o, y.to(cuda)).backward()
            torch.cuda.synchronize()
            return o
        _vs_emulation(mine, out, (e[4], e[6]), rerun, cancelled=["encoder.fc1.bias", "encoder.fc2.bias"])


@pytest.mark.timeout(420)
@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_resnet152_crnn_ucf_shape(cuda, dt):
    """The model the reference's ResNetCRNN actually instantiates: a ResNet-152 trunk
    (models/model_components.py:159-162) and a 3-layer LSTM (models/ResNetCRNN/UCF101_ResNetCRNN_refactored.py:49-51),
    6 clips of 16 x 3 x 224 x 224, against the CPU oracle: logits, loss, predictions and every gradient of the 50
    bottleneck blocks, the head and the decoder.  Same inputs and damping as the ResNet-50 test; three times the depth
    in front of the BatchNorm1d head: measured fp32 trunk 6e-3, head and decoder < 2e-5; bf16 trunk 0.67 against the
    fp32 oracle and 0.53 against the bf16-emulating oracle, where two runs of the device path on the same inputs
    already differ by 0.55 (fp32 atomics arriving in another order, amplified by 152 layers of batch statistics).
    Gradient bounds: fp32 2e-2; bf16 0.9 (trunk) / 0.12 (decoder), and the noise-floor bound against the emulation."""
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    cfg, sd, x, y, ref_out, ref_loss, ref_grads = _resnet_ref(False, "resnet152", 3, 6)
    f32 = dt == torch.float32
    mine = M.ResNetCRNNModel(cfg)
    mine.load_state_dict(sd, strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    mine.train()
    with _ordered_sums(f32):
        out = mine(x.to(cuda))
        loss = ops.cross_entropy(out, y.to(cuda))
        loss.backward()
        torch.cuda.synchronize()
    ops.check_lstm_error()
    # fp32: the device path runs with ordered sums (repeatable), and the head channels whose ReLU input lies within 1e-5
    # of zero in the oracle -- on this seed one of the 96 x 1024 inputs of the first ReLU -- are located in the oracle
    # and left out BY INDEX: that ReLU's decision, and with it the channel's row of the fc1 / bn1 gradients, is decided
    # by the last bit of the trunk's output (1.5e-5 when it agrees with the oracle's, 3.3e-3 when it does not).
    # Every other head and decoder gradient is held to 1e-4.
    _compare(mine, out, loss.item(), ref_out, ref_loss, ref_grads, dt, 2e-2 if f32 else 0.9,
             cancelled=["encoder.fc1.bias", "encoder.fc2.bias"],
             group_tol={"decoder.": 1e-4, "encoder.fc": 1e-4, "encoder.bn": 1e-4} if f32 else
             {"decoder.fc2": 3e-2, "decoder.": 0.12},
             exclude=_near_zero_relu_channels(False, "resnet152", 3, 6) if f32 else None)
    if not f32:
        e = _resnet_ref(True, "resnet152", 3, 6)

        def rerun():
            o = mine(x.to(cuda))
            ops.cross_entropy(o, y.to(cuda)).backward()
            torch.cuda.synchronize()
            return o
        _vs_emulation(mine, out, (e[4], e[6]), rerun, cancelled=["encoder.fc1.bias", "encoder.fc2.bias"])


@functools.lru_cache(maxsize=None)
def _swin_ref(emulate=False):
    from oracle import torch_oracle as O
    torch.manual_seed(12)
    ref = O.SwinTCGModel(O.ClusterSwinEncoder(embed_dim=96, k_tokens=8, swin_kwargs=NO_DROP), O.TCG_LSTM(96, 512, 2, 8, 4, 101, 0.0))
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(4, 16, 3, 224, 224, generator=g)
    y = torch.randint(0, 101, (4, 1), generator=g)
    ref.train()
    with _emulating(O, emulate):
        out = ref(x)
        loss = nn.CrossEntropyLoss()(out, y.squeeze(1))
        loss.backward()
    return ref.state_dict(), x, y, out.detach(), loss.item(), {k: p.grad for k, p in ref.named_parameters()}


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_cluster_swin_tcg_ucf_shape(cuda, dt):
    """No batch statistics anywhere in this model (LayerNorm trunk, no BatchNorm head): the best-conditioned of the
    three, and the tightest bounds: gradients 1e-4 in fp32, 5e-2 in bf16."""
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    from video_classification_amd import swin as S
    sd, x, y, ref_out, ref_loss, ref_grads = _swin_ref()
    mine = M.SwinTCGModel(S.ClusterSwinEncoder(embed_dim=96, k_tokens=8, swin_kwargs=NO_DROP), M.TCG_LSTM(96, 512, 2, 8, 4, 101, 0.0))
    mine.load_state_dict(sd, strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    mine.train()
    out = mine(x.to(cuda))
    loss = ops.cross_entropy(out, y.to(cuda))
    loss.backward()
    torch.cuda.synchronize()
    ops.check_lstm_error()
    # measured: fp32 worst 8e-6 (patch embedding); bf16 worst 1.5e-2 (a stage-4 bias table), decoder 7e-3
    _compare(mine, out, loss.item(), ref_out, ref_loss, ref_grads, dt, 1e-4 if dt == torch.float32 else 5e-2)
    if dt == torch.bfloat16:
        e = _swin_ref(True)

        def rerun():
            o = mine(x.to(cuda))
            ops.cross_entropy(o, y.to(cuda)).backward()
            torch.cuda.synchronize()
            return o
        _vs_emulation(mine, out, (e[3], e[5]), rerun, at_least=5e-2)  # measured 1.5e-2


@functools.lru_cache(maxsize=None)
def _swin_rnn_ref(emulate=False):
    from oracle import torch_oracle as O
    torch.manual_seed(14)
    # what the reference's factory builds for --model swintransformer-RNN (train.py:256-278 with the Config defaults
    # of models/swintransformer_RNN/UCF101_SwinCRNN_fixed.py:72-80 and create_model, :404-423)
    ref = O.SwinTransformerRNNModel(O.SwinTransformerEncoder(1024, 768, 0.0, 512, swin_kwargs=NO_DROP), O.DecoderRNN(512, 3, 512, 256, 0.0, 101))
    g = torch.Generator().manual_seed(777)
    B = 6
    x = torch.randn(B, 16, 3, 224, 224, generator=g)
    y = torch.randint(0, 101, (B, 1), generator=g)
    ref.train()
    with _emulating(O, emulate):
        out = ref(x)
        loss = nn.CrossEntropyLoss()(out, y.squeeze(1))
        loss.backward()
    bufs = {k: v.clone() for k, v in ref.named_buffers()}
    return ref.state_dict(), x, y, out.detach(), loss.item(), {k: p.grad for k, p in ref.named_parameters()}, bufs


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_swin_rnn_ucf_shape(cuda, dt):
    """configs[3], the assembled swintransformer-RNN model: Swin-T backbone -> BatchNorm1d head with per-time-step
    statistics over the 6 clips -> 3 x 512 LSTM -> FC, 16 x 3 x 224 x 224 clips.  Logits, loss, predictions, every
    gradient and the head's running statistics after the step against the CPU oracle."""
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    from video_classification_amd import swin as S
    mine = M.SwinTransformerRNNModel(S.SwinTransformerEncoder(1024, 768, 0.0, 512, swin_kwargs=NO_DROP), M.DecoderRNN(512, 3, 512, 256, 0.0, 101))
    sd, x, y, ref_out, ref_loss, ref_grads, ref_bufs = _swin_rnn_ref()
    # the oracle's state_dict is taken AFTER its training-mode forward: running statistics start from the defaults
    sd0 = {k: v for k, v in sd.items() if not ("running_" in k or "num_batches" in k)}
    missing, unexpected = mine.load_state_dict(sd0, strict=False)
    assert not unexpected and all(("running_" in k or "num_batches" in k) for k in missing)
    mine.to(cuda).set_compute_dtype(dt)
    mine.train()
    out = mine(x.to(cuda))
    loss = ops.cross_entropy(out, y.to(cuda))
    loss.backward()
    torch.cuda.synchronize()
    ops.check_lstm_error()
    f32 = dt == torch.float32
    # measured: fp32 backbone 1.2e-4, head and decoder < 1e-5; bf16 backbone 0.50, head 0.05-0.20 (BatchNorm1d over 6
    # samples per time step amplifies the rounding of everything in front of it, as in the ResNet test), decoder 1.2e-2
    _compare(mine, out, loss.item(), ref_out, ref_loss, ref_grads, dt, 2e-3 if f32 else 0.7,
             group_tol={"decoder.": 1e-4, "encoder.fc": 1e-4, "encoder.bn": 1e-4} if f32 else
             {"decoder.fc": 2e-2, "decoder.": 5e-2, "encoder.fc3": 5e-2},
             # constant shifts in front of a BatchNorm have a zero true gradient: the head's biases, and the
             # backbone's last LayerNorm bias (it passes the token mean and fc1 unchanged for every sample)
             cancelled=["encoder.fc1.bias", "encoder.fc2.bias", "encoder.swin.norm.bias"])
    mb = dict(mine.named_buffers())
    for k, v in ref_bufs.items():
        if k.endswith("num_batches_tracked"):
            assert int(mb[k]) == int(v) == 16, k   # one update per time step
        elif "running_" in k:
            assert G.max_rel(mb[k].cpu(), v) < (1e-4 if f32 else 2e-2), k
    if not f32:   # after the buffers: the second pass below updates the running statistics again
        e = _swin_rnn_ref(True)

        def rerun():
            o = mine(x.to(cuda))
            ops.cross_entropy(o, y.to(cuda)).backward()
            torch.cuda.synchronize()
            return o
        # BatchNorm1d over 6 samples amplifies the emulation's ~1e-3 infidelity as it amplifies bf16 rounding itself
        # (fp32: 1e-7 -> 1.2e-4 in this model): no tighter than the bound against the fp32 oracle (measured 0.63)
        _vs_emulation(mine, out, (e[3], e[5]), rerun, at_least=0.9,
                      cancelled=["encoder.fc1.bias", "encoder.fc2.bias", "encoder.swin.norm.bias"])


@functools.lru_cache(maxsize=None)
def _conv3d_ref(emulate=False):
    from oracle import torch_oracle as O
    torch.manual_seed(13)
    ref = O.Conv3DModel(G.conv3d_config())
    g = torch.Generator().manual_seed(999)
    x = torch.randn(4, 1, 16, 112, 112, generator=g)
    y = torch.randint(0, 101, (4, 1), generator=g)
    ref.train()
    with _emulating(O, emulate):
        out = ref(x)
        loss = nn.CrossEntropyLoss()(out, y.squeeze(1))
        loss.backward()
    return ref.state_dict(), x, y, out.detach(), loss.item(), {k: p.grad for k, p in ref.named_parameters()}


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_conv3d_native_shape(cuda, dt):
    """Gradient bounds 1e-2 in fp32, 0.5 in bf16 for the convolution stack; 1e-5 / 0.12 for the classifier."""
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    sd, x, y, ref_out, ref_loss, ref_grads = _conv3d_ref()
    mine = M.Conv3DModel(G.conv3d_config())
    mine.load_state_dict(sd, strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    mine.train()
    out = mine(x.to(cuda))
    loss = ops.cross_entropy(out, y.to(cuda))
    loss.backward()
    torch.cuda.synchronize()
    # The global average pool hands every position of a sample the same gradient, and each BatchNorm backward
    # removes its mean: heavy cancellation, so rounding is amplified layer by layer towards the input (measured
    # fp32: 6e-7 at the classifier, 8e-4 at the last convolution, 3e-3 at the first; bf16: 3e-3, 8e-2, 0.22).
    f32 = dt == torch.float32
    _compare(mine, out, loss.item(), ref_out, ref_loss, ref_grads, dt, 1e-2 if f32 else 0.5,
             cancelled=[f"conv3d.{i}.bias" for i in (0, 4, 8, 12)],
             group_tol={"fc.": 1e-5} if f32 else {"fc.4": 2e-2, "fc.": 0.12})
    if not f32:
        e = _conv3d_ref(True)

        def rerun():
            o = mine(x.to(cuda))
            ops.cross_entropy(o, y.to(cuda)).backward()
            torch.cuda.synchronize()
            return o
        # measured against the emulation: 0.054-0.076 at the first convolution in five runs, whose own repeats differ by
        # 0.02-0.064; the stated bound keeps a run with a small draw of that spread from failing
        _vs_emulation(mine, out, (e[3], e[5]), rerun, cancelled=[f"conv3d.{i}.bias" for i in (0, 4, 8, 12)],
                      at_least=0.1)
