"""The bench workload itself (BASELINE.json configs[1]: ResNetCRNN, 32 clips x 16 x 3 x 224 x 224), where the CPU
oracle would take minutes: size-independent properties and the two device paths against each other.

  * the bf16 MFMA path against the fp32 path of the same library (plain-FMA kernels, exact arithmetic order aside):
    logits within the stated bf16 output bound, same loss to 1e-2, predictions equal wherever the fp32 path's own
    top-2 margin exceeds the bound;
  * three optimizer steps through the phased hipGraph executor give the eager steps' loss sequence.
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
    bound = 4e-2 * ref.abs().max().item()
    assert (got - ref).abs().max().item() < bound
    assert abs(losses[torch.float32] - losses[torch.bfloat16]) < 1e-2
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * bound
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
