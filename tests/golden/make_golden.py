#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own classes.

Run in the build container only (the reference checkout is at /root/reference there and does
not travel to the GPU box):

    python tests/golden/make_golden.py

`utils/common/model_components.py` imports torchvision at module level and the image has no
torchvision, so an empty in-process stub module is registered under that name before the import
(nothing is installed or fetched).  ResCNNEncoder and ClusterSwinEncoder fetch a third-party network at
construction (torchvision resnet152, timm swin_tiny): for them small stand-in networks
(tests/golden_util.py: resnet_standin, swin_standin) are registered where the reference looks the real ones up,
and the reference's own constructor, head, per-frame loop and token glue run unchanged around them.  The VGG
branch of EncoderCNN is not covered.

Each fixture is an .npz of plain arrays: inputs, the initial state_dict ("p/<key>"), outputs,
gradients ("g/<key>"), post-step buffers / parameters ("a/<key>").
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True  # nothing is written into the reference checkout
import zlib

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("VC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))                    # tests/ (golden_util)
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))   # repository root (oracle, for the stand-ins)


def _import_reference():
    for name in ("torchvision", "torchvision.models", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, os.path.join(REF, "utils"))
    sys.path.insert(0, REF)
    import common.model_components as mc  # noqa: E402
    import common.training_utils as tu  # noqa: E402

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    conv3d = load("models/Conv3D/UCF101_3DCNN_refactored.py", "ref_conv3d")
    crnn = load("models/CRNN/UCF101_CRNN_refactored.py", "ref_crnn")
    return mc, tu, conv3d, crnn


def _sd(prefix, module):
    return {f"{prefix}/{k}": v.detach().clone().numpy() for k, v in module.state_dict().items()}


def _grads(module):
    return {f"g/{k}": p.grad.detach().clone().numpy() for k, p in module.named_parameters() if p.grad is not None}


def _save(name, d):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in d.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays)")


def _component_case(name, module, inputs, eval_too=True):
    """train-mode forward + backward of sum(out * w), then an eval-mode forward."""
    d = {}
    d.update(_sd("p", module))
    for k, v in inputs.items():
        d[f"in/{k}"] = v.detach().numpy()
    xs = [v.clone().requires_grad_(v.is_floating_point()) for v in inputs.values()]
    module.train()
    out = module(*xs)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(99))
    d["loss_w"] = w.numpy()
    (out * w).sum().backward()
    d["out_train"] = out.detach().numpy()
    d.update(_grads(module))
    for k, x in zip(inputs.keys(), xs):
        if x.grad is not None:
            d[f"gin/{k}"] = x.grad.numpy()
    bufs = {k for k, _ in module.named_buffers()}
    d.update({k: v for k, v in _sd("a", module).items() if k[2:] in bufs})  # parameters are unchanged here
    if eval_too:
        module.eval()
        with torch.no_grad():
            d["out_eval"] = module(*[v for v in inputs.values()]).numpy()
    _save(name, d)


def main():
    torch.set_num_threads(8)
    mc, tu, conv3d, crnn = _import_reference()
    g = torch.Generator()
    only = set(sys.argv[1:])  # fixture names to (re)generate; none = all

    if not only or "rescnn_standin" in only:
        # 8. ResCNNEncoder: the reference's constructor (children()[:-1] wrap, fc.in_features), its
        #    fc1-bn1-relu-fc2-bn2-relu-dropout-fc3 head and per-frame loop around a stand-in trunk
        import golden_util as GU
        sys.modules["torchvision.models"].resnet152 = GU.resnet_standin
        torch.manual_seed(0)
        enc = mc.ResCNNEncoder(fc_hidden1=64, fc_hidden2=48, drop_p=0.0, CNN_embed_dim=32, use_pretrained=False)
        with torch.no_grad():  # residual branches damped and non-trivial BatchNorm affines, as after training
            for n, p in enc.named_parameters():
                if n.endswith("bn3.weight"):
                    p.fill_(0.25)
                elif ".bn" in n or n.startswith("resnet.1."):
                    p.add_(0.1 * torch.randn(p.shape, generator=g.manual_seed(zlib.crc32(n.encode()) % 1000)))
        _component_case("rescnn_standin", enc, {"x": torch.randn(6, 3, 3, 64, 64, generator=g.manual_seed(11))})

    if not only or "cluster_swin_standin" in only:
        # 9. ClusterSwinEncoder: the reference's constructor (backbone probe, proj) and forward (per-frame loop,
        #    projection, k repeated "cluster tokens", stacking) around a stand-in backbone
        import golden_util as GU
        mc.create_model = GU.swin_standin
        torch.manual_seed(0)
        enc = mc.ClusterSwinEncoder(backbone="swin_tiny_patch4_window7_224", embed_dim=24, k_tokens=8)
        x = torch.randn(2, 4, 3, 56, 56, generator=g.manual_seed(12))
        d = {"in/x": x.numpy()}
        d.update(_sd("p", enc))
        enc.train()
        gg, cc = enc(x)
        wg = torch.randn(gg.shape, generator=g.manual_seed(13))
        wc = torch.randn(cc.shape, generator=g.manual_seed(14))
        ((gg * wg).sum() + (cc * wc).sum()).backward()
        d.update({"loss_wg": wg.numpy(), "loss_wc": wc.numpy(), "out_g": gg.detach().numpy(), "out_c": cc.detach().numpy()})
        d.update(_grads(enc))
        _save("cluster_swin_standin", d)
    if not only or "image_density_fusion" in only:
        # 10. ImageDensityFusion (models/swintransformer_RNN/image_density_fusion.py:131): channel / spatial attention
        #     block, the four-branch multi-scale block and the pooled 1x1 head.  The file imports cv2 (absent here) at
        #     module level without using it in these classes: an empty stub module stands in for it.
        if "cv2" not in sys.modules:
            sys.modules["cv2"] = types.ModuleType("cv2")
        spec = importlib.util.spec_from_file_location(
            "ref_idf", os.path.join(REF, "models/swintransformer_RNN/image_density_fusion.py"))
        idf = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(idf)
        torch.manual_seed(0)
        m = idf.create_image_density_fusion(input_channels=3, density_channels=1, feature_channels=32, output_channels=64)
        with torch.no_grad():  # non-trivial BatchNorm affines, as after training
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g.manual_seed(zlib.crc32(n.encode()) % 1000)))
        _component_case("image_density_fusion", m, {"img": torch.randn(4, 3, 16, 16, generator=g.manual_seed(21)),
                                                    "den": torch.rand(4, 1, 16, 16, generator=g.manual_seed(22))})
        # the density map at another resolution (bilinear resize inside forward) and the [B, T] wrapper
        torch.manual_seed(0)
        m = idf.create_image_density_fusion(input_channels=3, density_channels=1, feature_channels=32, output_channels=64)
        x = torch.randn(2, 2, 3, 16, 16, generator=g.manual_seed(23))
        den = torch.rand(2, 2, 1, 8, 8, generator=g.manual_seed(24))
        d = {"in/x": x.numpy(), "in/den": den.numpy()}
        d.update(_sd("p", m))
        m.train()
        out = m.forward_batch(x, den)
        w = torch.randn(out.shape, generator=g.manual_seed(25))
        (out * w).sum().backward()
        d.update({"loss_w": w.numpy(), "out_train": out.detach().numpy()})
        d.update(_grads(m))
        _save("image_density_fusion_batch", d)
    if only:
        return

    # 1. EncoderCNN tiny: pins per-time-step BatchNorm statistics and the T running-stat updates
    torch.manual_seed(0)
    enc = mc.EncoderCNN(img_x=64, img_y=64, fc_hidden1=64, fc_hidden2=48, drop_p=0.0, CNN_embed_dim=32)
    _component_case("encoder_cnn_tiny", enc, {"x": torch.randn(2, 3, 3, 64, 64, generator=g.manual_seed(1))})

    # 2. DecoderRNN tiny
    torch.manual_seed(0)
    dec = mc.DecoderRNN(CNN_embed_dim=32, h_RNN_layers=3, h_RNN=32, h_FC_dim=16, drop_p=0.0, num_classes=101)
    _component_case("decoder_rnn_tiny", dec, {"x": torch.randn(2, 5, 32, generator=g.manual_seed(2))})

    # 3. TCG_LSTM tiny (drop=0 so train mode is deterministic); c is an independent tensor here,
    #    not the encoder's repeat, so the mean over K is really exercised
    torch.manual_seed(0)
    tcg = mc.TCG_LSTM(embed_dim=24, hidden=32, layers=2, k_tokens=8, down=4, num_classes=101, drop=0.0)
    _component_case("tcg_lstm_tiny", tcg, {"g": torch.randn(2, 16, 24, generator=g.manual_seed(3)),
                                           "c": torch.randn(2, 16, 8, 24, generator=g.manual_seed(4))})

    # 4. Conv3DModel tiny (dropout 0)
    cfg3 = conv3d.Config()
    cfg3.dropout_p = 0.0
    cfg3.num_classes = 101
    torch.manual_seed(0)
    m3 = conv3d.Conv3DModel(cfg3)
    _component_case("conv3d_tiny", m3, {"x": torch.randn(2, 1, 4, 16, 16, generator=g.manual_seed(5))})

    # 5. Swin encoder head + per-frame loop with an injected stand-in backbone (the reference's forward
    #    runs unchanged; only `self.swin`, which needs timm weights, is replaced by a linear map to 768)
    torch.manual_seed(0)
    sw = mc.SwinTransformerEncoder(fc_hidden1=64, fc_hidden2=48, drop_p=0.0, CNN_embed_dim=32, use_pretrained=False)
    sw.swin = nn.Sequential(nn.Flatten(), nn.Linear(3 * 8 * 8, 768))
    _component_case("swin_head_tiny", sw, {"x": torch.randn(4, 3, 3, 8, 8, generator=g.manual_seed(6))})

    # 6. CRNN tiny, one full training batch through the reference's train_epoch (CE + Adam(wd 1e-4)),
    #    then validate_epoch on the same batch
    cfg = crnn.Config()
    cfg.img_x = cfg.img_y = 64
    cfg.CNN_fc_hidden1, cfg.CNN_fc_hidden2, cfg.CNN_embed_dim = 64, 48, 32
    cfg.RNN_hidden_layers, cfg.RNN_hidden_nodes, cfg.RNN_FC_dim = 3, 32, 16
    cfg.dropout_p = 0.0
    torch.manual_seed(0)
    model = crnn.CRNNModel(cfg)
    x = torch.randn(4, 3, 3, 64, 64, generator=g.manual_seed(7))
    y = torch.randint(0, 101, (4, 1), generator=g.manual_seed(8))
    d = {"in/x": x.numpy(), "in/y": y.numpy()}
    d.update(_sd("p", model))
    captured = []
    hook = model.register_forward_hook(lambda m, i, o: captured.append(o.detach().clone()))
    opt = tu.get_optimizer(model, "adam", lr=1e-3, weight_decay=1e-4)
    crit = nn.CrossEntropyLoss()
    loss, acc = tu.train_epoch(model, [(x, y)], crit, opt, torch.device("cpu"), 1)
    d["train_loss"] = np.float64(loss)
    d["train_acc"] = np.float64(acc)
    d["train_logits"] = captured[0].numpy()
    d.update(_grads(model))
    d.update(_sd("a", model))
    vloss, vacc = tu.validate_epoch(model, [(x, y)], crit, torch.device("cpu"), 1)
    d["val_loss"] = np.float64(vloss)
    d["val_acc"] = np.float64(vacc)
    d["val_logits"] = captured[1].numpy()
    hook.remove()
    _save("crnn_tiny_step", d)

    # 7. CRNN at the full configs[0] shape (B=4, T=16, 224x224, 1024/768/512, 3x512): weights come
    #    from manual_seed(0) default init (too large to store); per-tensor checksums pin that the
    #    regenerated weights are the same, and only small results are kept
    cfg = crnn.Config()
    torch.manual_seed(0)
    model = crnn.CRNNModel(cfg)
    d = {f"psum/{k}": np.float64(v.double().abs().sum().item()) for k, v in model.state_dict().items()}
    x = torch.randn(4, 16, 3, 224, 224, generator=g.manual_seed(1234))
    y = torch.randint(0, 101, (4, 1), generator=g.manual_seed(4321))
    d["in/y"] = y.numpy()
    model.train()
    out = model(x)
    loss = nn.CrossEntropyLoss()(out, y.squeeze())
    loss.backward()
    d["logits"] = out.detach().numpy()
    d["loss"] = np.float64(loss.item())
    d["argmax"] = out.argmax(1).numpy()
    for k, p in model.named_parameters():
        d[f"gnorm/{k}"] = np.float64(p.grad.double().norm().item())
    # a small slice of one gradient as a direct (not only norm) check
    d["g/decoder.fc2.weight"] = model.decoder.fc2.weight.grad.numpy()
    d["g/encoder.conv1.0.weight"] = model.encoder.conv1[0].weight.grad.numpy()
    model.eval()
    with torch.no_grad():
        d["logits_eval"] = model(x).numpy()
    _save("crnn_cfg0", d)


if __name__ == "__main__":
    main()
