"""``create_model(model_name, config)``: the reference's model factory (train.py:81-326) over the HIP modules.

Same model names, same YAML keys read, same defaults as the reference's per-model ``Config`` classes
(models/Conv3D/UCF101_3DCNN_refactored.py:36-55, models/CRNN/UCF101_CRNN_refactored.py:36-62,
models/ResNetCRNN/UCF101_ResNetCRNN_refactored.py:36-62, models/swintransformer_RNN/UCF101_SwinCRNN_fixed.py:60-145),
so a ``configs/*.yaml`` of the reference builds a model whose ``state_dict`` has the reference's keys, shapes and dtypes.
Keys the reference's factory does not read (``model.cnn_encoder``, ``model.num_layers`` for CRNN / ResNetCRNN,
``model.clusterSwin.depths`` ...) are not read here either.  ``config`` is the parsed YAML dictionary, or an object
carrying it as ``.config`` (the reference's ``ConfigLoader``).

Differences to know about: pretrained ImageNet weights are never downloaded (``checkpoint.pretrained`` is accepted and
ignored; load them with ``load_state_dict``), and ``ResNetCRNN_varylength`` is refused: the reference's own module for it
does not import (models/ResNetCRNN_varylength/UCF101_ResNetCRNN_varlen.py:13, :143).
"""
import types

from . import modules as M
from . import swin as S

MODEL_NAMES = ("Conv3D", "CRNN", "ResNetCRNN", "swintransformer-RNN", "clusterSwin_TCGlstm")


def _dict_of(config):
    d = getattr(config, "config", config)
    if not isinstance(d, dict):
        raise TypeError("config must be the parsed YAML dictionary or an object with a .config dictionary")
    return d


def create_model(model_name, config):
    cfg = _dict_of(config)
    data = cfg.get("data", {}) or {}
    model = cfg.get("model", {}) or {}
    ckpt = cfg.get("checkpoint", {}) or {}
    num_classes = int(data.get("num_classes", 101))

    if model_name == "Conv3D":  # train.py:86-117
        return M.Conv3DModel(types.SimpleNamespace(dropout_p=0.5, num_classes=num_classes))

    if model_name == "CRNN":  # train.py:119-160
        size = int(data.get("frame_size", 224))
        c = types.SimpleNamespace(img_x=size, img_y=size, CNN_fc_hidden1=1024, CNN_fc_hidden2=768, CNN_embed_dim=512,
                                  dropout_p=0.0, use_pretrained=bool(ckpt.get("pretrained", False)),
                                  RNN_hidden_layers=3, RNN_hidden_nodes=512, RNN_FC_dim=256, num_classes=num_classes)
        if c.use_pretrained:
            raise NotImplementedError("CRNN with checkpoint.pretrained selects the VGG16 branch of EncoderCNN "
                                      "(model_components.py:72-79), which needs torchvision's weights")
        return M.CRNNModel(c)

    if model_name == "ResNetCRNN":  # train.py:162-215
        c = types.SimpleNamespace(CNN_fc_hidden1=int(model.get("fc_hidden1", 1024)),
                                  CNN_fc_hidden2=int(model.get("fc_hidden2", 768)),
                                  CNN_embed_dim=int(model.get("embed_dim", 512)),
                                  dropout_p=float(model.get("dropout", 0.0)),
                                  use_pretrained=bool(ckpt.get("pretrained", True)),
                                  RNN_hidden_layers=int(model.get("hidden_layers", 3)),
                                  RNN_hidden_nodes=int(model.get("hidden_nodes", 512)),
                                  RNN_FC_dim=int(model.get("fc_dim", 256)), num_classes=num_classes,
                                  arch="resnet152")  # model_components.py:159-162 builds resnet152 whatever the YAML says
        return M.ResNetCRNNModel(c)

    if model_name == "swintransformer-RNN":  # train.py:256-278 with Config / create_model of UCF101_SwinCRNN_fixed.py
        enc = S.SwinTransformerEncoder(fc_hidden1=1024, fc_hidden2=768, drop_p=0.0, CNN_embed_dim=512,
                                       swin_model_name="swin_tiny_patch4_window7_224",
                                       use_pretrained=bool(ckpt.get("pretrained", True)))
        dec = M.DecoderRNN(CNN_embed_dim=512, h_RNN_layers=3, h_RNN=512, h_FC_dim=256, drop_p=0.0,
                           num_classes=num_classes)
        return M.SwinTransformerRNNModel(enc, dec)

    if model_name == "clusterSwin_TCGlstm":  # train.py:280-323
        cs = model.get("clusterSwin", {}) or {}
        embed = int(cs.get("embed_dim", 512))
        enc = S.ClusterSwinEncoder(backbone="swin_tiny_patch4_window7_224", embed_dim=embed, k_tokens=8)
        dec = M.TCG_LSTM(embed_dim=embed, hidden=int(model.get("hidden_size", 512)),
                         layers=int(model.get("num_layers", 3)), k_tokens=8, down=4, num_classes=num_classes,
                         drop=float(model.get("dropout", 0.0)))
        return M.SwinTCGModel(enc, dec)

    if model_name == "ResNetCRNN_varylength":
        raise NotImplementedError("ResNetCRNN_varylength: the reference's module for it does not import "
                                  "(UCF101_ResNetCRNN_varlen.py:13, :143), so there is no behaviour to reproduce")
    raise ValueError(f"Unsupported model: {model_name}")
