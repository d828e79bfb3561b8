"""Helpers for reading the golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py
from the reference's own classes)."""
import os
import types

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def group(d, prefix):
    """Sub-dictionary of keys 'prefix/<k>' as torch tensors."""
    n = len(prefix) + 1
    return {k[n:]: torch.from_numpy(np.array(v)) for k, v in d.items() if k.startswith(prefix + "/")}


def tiny_crnn_config():
    return types.SimpleNamespace(img_x=64, img_y=64, CNN_fc_hidden1=64, CNN_fc_hidden2=48, CNN_embed_dim=32,
                                 dropout_p=0.0, use_pretrained=False, RNN_hidden_layers=3, RNN_hidden_nodes=32,
                                 RNN_FC_dim=16, num_classes=101)


def full_crnn_config():
    return types.SimpleNamespace(img_x=224, img_y=224, CNN_fc_hidden1=1024, CNN_fc_hidden2=768, CNN_embed_dim=512,
                                 dropout_p=0.0, use_pretrained=False, RNN_hidden_layers=3, RNN_hidden_nodes=512,
                                 RNN_FC_dim=256, num_classes=101)


def conv3d_config():
    return types.SimpleNamespace(dropout_p=0.0, num_classes=101)


# ---- stand-ins for the third-party networks the reference fetches at construction -------------------------------
# make_golden.py registers them where the reference looks its backbones up (torchvision.models.resnet152,
# model_components.create_model), so that the reference's OWN ResCNNEncoder / ClusterSwinEncoder code runs
# unchanged around them; the tests build the same stand-ins to know what the fixtures' keys mean.
RESNET_STANDIN = dict(arch="tiny", width=8)      # bottleneck trunk [1, 1, 1, 1], 8 stem channels, 256 features
SWIN_STANDIN = dict(img=56, patch=4, dim=32, depths=(2, 2), heads=(1, 2), ws=7, drop_path_rate=0.0)


def resnet_standin(weights=None):
    """A module with torchvision's ResNet child layout (conv1, bn1, relu, maxpool, layer1-4, avgpool, fc): the
    reference wraps list(children())[:-1] and reads fc.in_features (model_components.py:164-166)."""
    import torch.nn as nn
    from oracle import torch_oracle as O
    O.RESNET_LAYERS.setdefault("tiny", (1, 1, 1, 1))
    trunk = O.make_resnet_trunk(**RESNET_STANDIN)
    m = nn.Module()
    for name, child in zip(("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool"), trunk):
        m.add_module(name, child)
    m.add_module("fc", nn.Linear(32 * RESNET_STANDIN["width"], 10))
    return m


def swin_standin(name="swin_tiny_patch4_window7_224", pretrained=True, num_classes=0):
    """Stands in for timm.create_model(name, pretrained=True, num_classes=0): a two-stage Swin of the oracle's
    restatement.  The reference's constructor probes its backbone with one 224 x 224 frame
    (model_components.py:419-422); the stand-in looks at that frame's top-left 56 x 56 corner."""
    from oracle import torch_oracle as O

    class Backbone(O.SwinT):
        def forward(self, x):
            n = SWIN_STANDIN["img"]
            return super().forward(x[..., :n, :n])
    return Backbone(**SWIN_STANDIN)


def max_rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def l2_rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def check_grads(named_params, grefs, tol, cancelled=()):
    """Compare parameter gradients with the reference's.

    tol bounds the L2 error relative to the tensor's norm and, on the fp32 path (tol <= 1e-2), also the
    max-norm error relative to the tensor's largest magnitude at 3x tol.  `cancelled` lists biases that feed straight into a
    BatchNorm: their true gradient is exactly zero (the mean subtraction cancels them), both sides only
    hold rounding noise there, so they are bounded against the gradient scale of the layer's weight."""
    for k, gref in grefs.items():
        g = named_params[k].grad
        assert g is not None, k
        g = g.detach().cpu()
        if any(k.endswith(c) for c in cancelled):
            wkey = k[:-len("bias")] + "weight"
            bound = tol * torch.as_tensor(grefs[wkey]).abs().max().item()
            assert g.abs().max().item() <= bound, (k, g.abs().max().item(), bound)
            continue
        assert l2_rel(g, gref) < tol, (k, "l2", l2_rel(g, gref))
        if tol <= 1e-2:  # the max-norm only means something on the exact (fp32) path
            assert max_rel(g, gref) < 3 * tol, (k, "max", max_rel(g, gref))


def report_against_noise_floor(title, err, floor):
    w, f = max(err, key=err.get), max(floor, key=floor.get)
    med = lambda d: sorted(d.values())[len(d) // 2]  # noqa: E731
    print(f"\n[{title}] worst {w} {err[w]:.3e} (run-to-run worst {f} {floor[f]:.3e}), median {med(err):.3e} "
          f"(run-to-run {med(floor):.3e})")


def assert_within_noise_floor(err, floor, factor=2.0, slack=5e-3, at_least=0.0):
    """err[k]: relative L2 distance of tensor k to the bf16-emulating oracle; floor[k]: distance between two runs of the
    HIP path on identical inputs (its own noise floor: fp32 atomics arrive in a different order, and bf16 rounding
    amplifies any difference up to the rounding noise).  The oracle may be further from a run than the run's repetition
    by `factor` (worst tensor and median over tensors) plus `slack`.  `at_least`: bound for paths that repeat
    (nearly) bit for bit, where the floor says nothing and the distance is the emulation's own infidelity."""
    med = lambda d: sorted(d.values())[len(d) // 2]  # noqa: E731
    assert max(err.values()) <= max(factor * max(floor.values()) + slack, at_least), (
        max(err, key=err.get), max(err.values()), max(floor.values()))
    assert med(err) <= max(factor * med(floor) + slack, at_least), (med(err), med(floor))
