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


def max_rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
