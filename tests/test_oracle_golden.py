"""Pins the CPU oracle (oracle/torch_oracle.py) against golden vectors produced by the reference's
own classes (tests/golden/make_golden.py).  fp32 on CPU on both sides, so the bar is 1e-5 relative
to each tensor's largest magnitude (different operator batching, same arithmetic)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import golden_util as G
from oracle import torch_oracle as O

TOL = 1e-5


def _run_component(name, module, input_names):
    d = G.load(name)
    module.load_state_dict(G.group(d, "p"), strict=True)
    ins = [torch.from_numpy(d[f"in/{k}"]).clone().requires_grad_(True) for k in input_names]
    module.train()
    out = module(*ins)
    (out * torch.from_numpy(d["loss_w"])).sum().backward()
    assert G.max_rel(out.detach(), d["out_train"]) < TOL
    for k, gref in G.group(d, "g").items():
        p = dict(module.named_parameters())[k]
        assert G.max_rel(p.grad, gref) < 5 * TOL, k
    for k, x in zip(input_names, ins):
        assert G.max_rel(x.grad, d[f"gin/{k}"]) < 5 * TOL, k
    bufs = dict(module.named_buffers())
    for k, bref in G.group(d, "a").items():
        assert G.max_rel(bufs[k].double(), bref.double()) < TOL, k
    module.eval()
    with torch.no_grad():
        oe = module(*[torch.from_numpy(d[f"in/{k}"]) for k in input_names])
    assert G.max_rel(oe, d["out_eval"]) < TOL


def test_encoder_cnn():
    _run_component("encoder_cnn_tiny", O.EncoderCNN(64, 64, 64, 48, 0.0, 32), ["x"])


def test_decoder_rnn():
    _run_component("decoder_rnn_tiny", O.DecoderRNN(32, 3, 32, 16, 0.0, 101), ["x"])


def test_tcg_lstm():
    _run_component("tcg_lstm_tiny", O.TCG_LSTM(24, 32, 2, 8, 4, 101, 0.0), ["g", "c"])


def test_conv3d():
    _run_component("conv3d_tiny", O.Conv3DModel(G.conv3d_config()), ["x"])


def test_swin_head_and_frame_loop():
    """The reference's SwinTransformerEncoder.forward with a stand-in backbone (see make_golden.py)."""
    enc = O.SwinTransformerEncoder(64, 48, 0.0, 32, swin_kwargs=dict(drop_path_rate=0.0))
    enc.swin = nn.Sequential(nn.Flatten(), nn.Linear(3 * 8 * 8, 768))
    _run_component("swin_head_tiny", enc, ["x"])


def test_rescnn_encoder_around_standin_trunk():
    """The reference's ResCNNEncoder (constructor wrap of children()[:-1], head, per-frame loop) around the stand-in
    trunk of golden_util.resnet_standin: pins model_components.py:164-191."""
    O.RESNET_LAYERS.setdefault("tiny", (1, 1, 1, 1))
    _run_component("rescnn_standin", O.ResCNNEncoder(64, 48, 0.0, 32, use_pretrained=False, **G.RESNET_STANDIN), ["x"])


def test_cluster_swin_encoder_around_standin_backbone():
    """The reference's ClusterSwinEncoder.forward (per-frame loop, projection, k repeated tokens, stacking:
    model_components.py:452-475) around the stand-in backbone of golden_util.swin_standin."""
    d = G.load("cluster_swin_standin")
    enc = O.ClusterSwinEncoder(embed_dim=24, k_tokens=8, swin_kwargs=G.SWIN_STANDIN)
    enc.load_state_dict(G.group(d, "p"), strict=True)
    enc.train()
    g, c = enc(torch.from_numpy(d["in/x"]))
    ((g * torch.from_numpy(d["loss_wg"])).sum() + (c * torch.from_numpy(d["loss_wc"])).sum()).backward()
    assert tuple(c.shape) == (2, 4, 8, 24)
    assert G.max_rel(g.detach(), d["out_g"]) < TOL and G.max_rel(c.detach(), d["out_c"]) < TOL
    params = dict(enc.named_parameters())
    grefs = G.group(d, "g")
    assert set(grefs) == set(params)
    for k, gref in grefs.items():
        assert G.max_rel(params[k].grad, gref) < 5 * TOL, k


def test_crnn_train_step_and_eval():
    d = G.load("crnn_tiny_step")
    model = O.CRNNModel(G.tiny_crnn_config())
    model.load_state_dict(G.group(d, "p"), strict=True)
    x, y = torch.from_numpy(d["in/x"]), torch.from_numpy(d["in/y"])
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = nn.CrossEntropyLoss()
    loss, logits, correct = O.train_step(model, x, y, crit, opt)
    assert abs(loss - float(d["train_loss"])) < 1e-5
    assert G.max_rel(logits, d["train_logits"]) < TOL
    assert 100.0 * correct / 4 == float(d["train_acc"])
    for k, gref in G.group(d, "g").items():
        assert G.max_rel(dict(model.named_parameters())[k].grad, gref) < 5 * TOL, k
    sd = model.state_dict()
    for k, aref in G.group(d, "a").items():
        # Adam's first step moves every weight by ~lr regardless of gradient scale: compare absolutely
        assert (sd[k].double() - aref.double()).abs().max().item() < 2e-5, k
    vloss, vlogits, vcorrect = O.eval_step(model, x, y, crit)
    assert abs(vloss - float(d["val_loss"])) < 1e-4
    assert G.max_rel(vlogits, d["val_logits"]) < 1e-4
    assert torch.equal(vlogits.argmax(1), torch.from_numpy(d["val_logits"]).argmax(1))


@pytest.mark.timeout(300)
def test_crnn_full_config0():
    """configs[0]: CRNN fwd/bwd, batch 4 x 16 frames x 224^2, 101 classes, fp32 on CPU."""
    d = G.load("crnn_cfg0")
    torch.manual_seed(0)
    model = O.CRNNModel(G.full_crnn_config())
    for k, v in model.state_dict().items():  # same default-init stream as the reference's classes
        assert abs(v.double().abs().sum().item() - float(d[f"psum/{k}"])) <= 1e-9 * max(1.0, float(d[f"psum/{k}"])), k
    x = torch.randn(4, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1234))
    y = torch.from_numpy(d["in/y"])
    model.train()
    out = model(x)
    loss = nn.CrossEntropyLoss()(out, y.squeeze())
    loss.backward()
    assert G.max_rel(out.detach(), d["logits"]) < TOL
    assert abs(loss.item() - float(d["loss"])) < 1e-5
    assert np.array_equal(out.argmax(1).numpy(), d["argmax"])
    for k, p in model.named_parameters():
        ref = float(d[f"gnorm/{k}"])
        assert abs(p.grad.double().norm().item() - ref) <= 1e-4 * ref + 1e-12, k
    assert G.max_rel(model.decoder.fc2.weight.grad, d["g/decoder.fc2.weight"]) < 5 * TOL
    assert G.max_rel(model.encoder.conv1[0].weight.grad, d["g/encoder.conv1.0.weight"]) < 5 * TOL


def test_image_density_fusion():
    """The fusion CNN against the reference's own class (models/swintransformer_RNN/image_density_fusion.py:131)."""
    _run_component("image_density_fusion", O.ImageDensityFusion(3, 1, 32, 64), ["img", "den"])


def test_image_density_fusion_batch_and_resized_density():
    d = G.load("image_density_fusion_batch")
    m = O.ImageDensityFusion(3, 1, 32, 64)
    m.load_state_dict(G.group(d, "p"), strict=True)
    m.train()
    out = m.forward_batch(torch.from_numpy(d["in/x"]), torch.from_numpy(d["in/den"]))
    (out * torch.from_numpy(d["loss_w"])).sum().backward()
    assert G.max_rel(out.detach(), d["out_train"]) < 1e-5
    for k, p in m.named_parameters():
        assert G.max_rel(p.grad, d[f"g/{k}"]) < 1e-4, k
