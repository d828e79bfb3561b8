"""The bench workload itself (BASELINE.json configs[1]: ResNetCRNN, 32 clips x 16 x 3 x 224 x 224), where the CPU
oracle would take minutes: size-independent properties and the two device paths against each other.

  * the bf16 MFMA path against the fp32 path of the same library (plain-FMA kernels, exact arithmetic order aside):
    logits within the stated bf16 output bound, same loss to 1e-2, predictions equal wherever the fp32 path's own
    top-2 margin exceeds the bound;
  * three optimizer steps through the phased hipGraph executor give the eager steps' loss sequence;
  * the batch of the benchmark against the CPU oracle (about 20 s of host time per oracle pass on the GPU box: the
    oracle keeps ~70 GB of fp32 activations, more than the build container has): fp32 path with ordered sums against
    the fp32 oracle, bf16 path against the oracle with bf16 storage emulation.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cuda, dt, seed=0):
    import types
    from video_classification_amd import modules as M
    torch.manual_seed(seed)
    cfg = types.SimpleNamespace(CNN_fc_hidden1=1024, CNN_fc_hidden2=768, CNN_embed_dim=512, dropout_p=0.0,
                                use_pretrained=False, RNN_hidden_layers=2, RNN_hidden_nodes=512, RNN_FC_dim=256,
                                num_classes=101, arch="resnet50")
    m = M.ResNetCRNNModel(cfg)
    with torch.no_grad():  # damped residual branches, as training leaves them (see test_gpu_fullsize)
        for n, p in m.named_parameters():
            if n.endswith("bn3.weight"):
                p.fill_(0.1)
    return m.to(cuda).set_compute_dtype(dt)


def _clips(cuda, B=32):
    g = torch.Generator().manual_seed(77)
    base = torch.randn(B, 1, 3, 28, 28, generator=g)
    x = torch.nn.functional.interpolate(base.view(B, 3, 28, 28), size=224, mode="bilinear").view(B, 1, 3, 224, 224)
    x = x + 0.25 * torch.randn(B, 16, 3, 28, 28, generator=g).repeat_interleave(8, -1).repeat_interleave(8, -2)
    y = torch.randint(0, 101, (B, 1), generator=g)
    return x.to(cuda), y.to(cuda)


@pytest.mark.timeout(600)
def test_bench_shape_bf16_against_fp32_device_path(cuda):
    from video_classification_amd import ops
    x, y = _clips(cuda)
    outs, losses = {}, {}
    sd = None
    for dt in (torch.float32, torch.bfloat16):
        m = _model(cuda, dt)
        if sd is None:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
        else:
            m.load_state_dict(sd)
        m.train()
        out = m(x)
        loss = ops.cross_entropy(out, y)
        loss.backward()
        torch.cuda.synchronize()
        ops.check_lstm_error()
        assert torch.isfinite(out).all()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters())
        outs[dt], losses[dt] = out.detach().float(), loss.item()
        del m
        torch.cuda.empty_cache()
    ref, got = outs[torch.float32], outs[torch.bfloat16]
    bound = 1e-2 * ref.abs().max().item()   # measured 3e-3 of the largest logit
    err = (got - ref).abs().max().item()
    assert err < bound, (err, bound)
    assert abs(losses[torch.float32] - losses[torch.bfloat16]) < 1e-2
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * bound
    print(f"\n  logits error {err / ref.abs().max().item():.2e} of the largest logit; predictions compared: "
          f"{int(clear.sum())} of {ref.shape[0]} rows")
    assert int(clear.sum()) >= ref.shape[0] // 2
    assert torch.equal(ref.argmax(1)[clear], got.argmax(1)[clear])


@pytest.mark.timeout(600)
def test_bench_shape_graph_step_equals_eager_step(cuda):
    """Three optimizer steps at the bench shape through the phased hipGraph executor and through eager launches from
    the same initial state: same losses up to the run-to-run noise of the bf16 path (fp32 atomics)."""
    from video_classification_amd import trainer as TR
    from video_classification_amd.graph import GraphedTrainStep
    x, y = _clips(cuda)
    losses = {}
    sd = None
    for mode in ("eager", "graph"):
        m = _model(cuda, torch.bfloat16)
        if sd is None:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
        else:
            m.load_state_dict(sd)
        opt = TR.get_optimizer(m, "adam", lr=1e-4, weight_decay=0.0)
        crit = TR.FusedCrossEntropyLoss()
        m.train()
        seq = []
        if mode == "eager":
            for _ in range(3 + 3):   # the graph executor runs 3 eager warm-up steps before capture
                opt.zero_grad()
                loss = crit(m(x), y)
                loss.backward()
                opt.step()
                seq.append(loss.item())
        else:
            step = GraphedTrainStep(m, opt, crit, x, y, warmup=3)
            seq = [None] * 3 + [float(step(x, y).detach()) for _ in range(3)]
        torch.cuda.synchronize()
        losses[mode] = seq[3:]
        del m, opt
        torch.cuda.empty_cache()
    for a, b in zip(losses["eager"], losses["graph"]):
        assert abs(a - b) < 2e-2 * max(1.0, abs(a)), (losses["eager"], losses["graph"])
    assert losses["graph"][-1] < losses["graph"][0] + 1e-3   # training on one batch does not go up


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_bench_batch_against_oracle(cuda, dt):
    """configs[1] at the batch bench.py runs (32 clips of 16 x 3 x 224 x 224, ResNet-50 trunk, 2 x 512 LSTM) against the
    CPU oracle with the same weights and clips (the step of training_utils.py:84-92: forward, mean cross-entropy,
    backward): logits, loss, predictions, every parameter gradient.  With 32 rows per BatchNorm group (instead of the 6 of
    tests/test_gpu_fullsize.py) the batch statistics are far better conditioned.
    fp32 path (ordered sums): logits 2e-4, trunk gradients 2e-2, head and decoder 1e-4 except the head channels whose
    ReLU input lies within 1e-5 of zero in the oracle (located there, left out by index).
    bf16 path against the bf16-emulating oracle: bounds from the measured errors, stated below."""
    import test_gpu_fullsize as FS
    from video_classification_amd import modules as M
    from video_classification_amd import ops
    f32 = dt == torch.float32
    cfg, sd, x, y, ref_out, ref_loss, ref_grads = FS._resnet_ref(not f32, "resnet50", 2, 32)
    mine = M.ResNetCRNNModel(cfg)
    mine.load_state_dict(sd, strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    mine.train()
    with FS._ordered_sums(f32):
        out = mine(x.to(cuda))
        loss = ops.cross_entropy(out, y.to(cuda))
        loss.backward()
        torch.cuda.synchronize()
    ops.check_lstm_error()
    if f32:
        FS._compare(mine, out, loss.item(), ref_out, ref_loss, ref_grads, dt, 2e-2,
                    cancelled=["encoder.fc1.bias", "encoder.fc2.bias"],
                    group_tol={"decoder.": 1e-4, "encoder.fc": 1e-4, "encoder.bn": 1e-4},
                    exclude=FS._near_zero_relu_channels(False, "resnet50", 2, 32))
    else:
        # measured at this batch (the per-group table the test prints): logits 3.0e-3 of the largest logit; gradients,
        # relative L2 against the emulating oracle: decoder.fc2 5e-3, decoder 5.7e-2, head 6e-2 .. 0.18, trunk 0.30 (0.40 at
        # 6 clips), while two runs of the device path on the same inputs differ by about as much (fp32 atomics in another
        # order, amplified by 50 layers of batch statistics in bf16): the second check holds the distance to the emulation
        # to twice that measured floor.  Bounds = about twice the measured values; predictions are compared on the rows
        # whose top-2 margin exceeds twice the OUTPUT bound of 6e-3, at least half the batch.
        FS._compare(mine, out, loss.item(), ref_out, ref_loss, ref_grads, dt, 0.6,
                    cancelled=["encoder.fc1.bias", "encoder.fc2.bias"],
                    group_tol={"decoder.fc2": 2e-2, "decoder.": 0.12, "encoder.fc": 0.35, "encoder.bn": 0.35},
                    tol_o=6e-3, min_rows=16)

        def rerun():
            o = mine(x.to(cuda))
            ops.cross_entropy(o, y.to(cuda)).backward()
            torch.cuda.synchronize()
            return o
        FS._vs_emulation(mine, out, (ref_out, ref_grads), rerun, cancelled=["encoder.fc1.bias", "encoder.fc2.bias"])
