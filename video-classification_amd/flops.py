"""Algorithmic forward FLOPs per clip of the hot-path models (2 x multiply-accumulates of the convolutions,
linear layers, LSTM gate products and attention products; biases, normalisation and activations not counted).
This is the figure the roofline fractions are computed from; a training step is counted as 3x forward.

    flops_per_clip(model, T=16, H=224, W=224)     any of the classes in modules.py / swin.py
"""
import torch.nn as nn

from . import modules as M
from . import swin as S


def _co(size, k, s, p):
    return (size + 2 * p - k) // s + 1


def conv2d_flops(conv, h, w):
    """(flops, out_h, out_w) of an nn.Conv2d on an h x w map."""
    kh, kw = conv.kernel_size
    ho, wo = _co(h, kh, conv.stride[0], conv.padding[0]), _co(w, kw, conv.stride[1], conv.padding[1])
    return 2.0 * ho * wo * conv.out_channels * (conv.in_channels // conv.groups) * kh * kw, ho, wo


def linear_flops(lin, rows=1):
    return 2.0 * rows * lin.in_features * lin.out_features


def lstm_flops(lstm, T):
    """nn.LSTM over T steps: per layer and direction 4H x (In + H) multiply-accumulates per step."""
    dirs = 2 if lstm.bidirectional else 1
    H, total = lstm.hidden_size, 0.0
    for l in range(lstm.num_layers):
        inp = lstm.input_size if l == 0 else H * dirs
        total += 2.0 * T * dirs * 4 * H * (inp + H)
    return total


def encoder_cnn_flops(m, H, W):
    """EncoderCNN, one frame."""
    total, h, w = 0.0, H, W
    for i in range(4):
        f, h, w = conv2d_flops(getattr(m, f"conv{i + 1}")[0], h, w)
        total += f
    return total + linear_flops(m.fc1) + linear_flops(m.fc2) + linear_flops(m.fc3)


def resnet_trunk_flops(trunk, H, W):
    """Bottleneck ResNet trunk (conv1 .. layer4), one frame."""
    f, h, w = conv2d_flops(trunk[0], H, W)
    total = f
    h, w = _co(h, 3, 2, 1), _co(w, 3, 2, 1)  # max-pool
    for li in range(4, 8):
        for blk in trunk[li]:
            if blk.downsample is not None:
                total += conv2d_flops(blk.downsample[0], h, w)[0]
            f1, h1, w1 = conv2d_flops(blk.conv1, h, w)
            f2, h2, w2 = conv2d_flops(blk.conv2, h1, w1)
            f3, h, w = conv2d_flops(blk.conv3, h2, w2)
            total += f1 + f2 + f3
    return total


def head_flops(m):
    """fc1 / fc2 / fc3 head shared by the ResNet and Swin encoders, one frame."""
    return linear_flops(m.fc1) + linear_flops(m.fc2) + linear_flops(m.fc3)


def swin_flops(sw, H, W):
    """SwinT backbone, one frame: patch embedding, per block QKV / attention (QK^T and PV over ws^2 tokens per
    window and head) / projection / MLP, patch-merging reductions."""
    f, h, w = conv2d_flops(sw.patch_embed.proj, H, W)
    total = f
    for stage in sw.layers:
        if not isinstance(stage.downsample, nn.Identity):
            h, w = h // 2, w // 2
            total += linear_flops(stage.downsample.reduction, h * w)
        for blk in stage.blocks:
            n = h * w
            total += linear_flops(blk.attn.qkv, n) + linear_flops(blk.attn.proj, n)
            total += linear_flops(blk.mlp.fc1, n) + linear_flops(blk.mlp.fc2, n)
            ws2 = blk.ws * blk.ws
            total += 2.0 * 2.0 * n * ws2 * blk.attn.proj.in_features   # QK^T and PV: per token ws^2 keys x dim
    return total


def decoder_rnn_flops(m, T):
    return lstm_flops(m.LSTM, T) + linear_flops(m.fc1) + linear_flops(m.fc2)


def tcg_lstm_flops(m, T):
    Tl = (T + m.down - 1) // m.down
    return (lstm_flops(m.lstm_short, T) + lstm_flops(m.lstm_long, Tl) + linear_flops(m.gate[0], T) +
            linear_flops(m.fc[0]) + linear_flops(m.fc[3]))


def conv3d_model_flops(m, T, H, W):
    total, d, h, w = 0.0, T, H, W
    for i in range(4):
        cv = m.conv3d[4 * i]
        k = cv.kernel_size
        total += 2.0 * d * h * w * cv.out_channels * cv.in_channels * k[0] * k[1] * k[2]  # padding 1: same size
        h, w = h // 2, w // 2
    return total + linear_flops(m.fc[1]) + linear_flops(m.fc[4])


def flops_per_clip(model, T=16, H=224, W=224):
    """Forward FLOPs of one clip of T frames (H x W) through `model`."""
    if isinstance(model, M.EncoderCNN):
        return T * encoder_cnn_flops(model, H, W)
    if isinstance(model, M.ResCNNEncoder):
        return T * (resnet_trunk_flops(model.resnet, H, W) + head_flops(model))
    if isinstance(model, S.SwinTransformerEncoder):
        return T * (swin_flops(model.swin, H, W) + head_flops(model))
    if isinstance(model, S.ClusterSwinEncoder):
        return T * (swin_flops(model.backbone, H, W) + linear_flops(model.proj))
    if isinstance(model, M.DecoderRNN):
        return decoder_rnn_flops(model, T)
    if isinstance(model, M.TCG_LSTM):
        return tcg_lstm_flops(model, T)
    if isinstance(model, M.Conv3DModel):
        return conv3d_model_flops(model, T, H, W)
    if hasattr(model, "encoder") and hasattr(model, "decoder"):
        return flops_per_clip(model.encoder, T, H, W) + flops_per_clip(model.decoder, T, H, W)
    raise TypeError(f"no FLOP model for {type(model).__name__}")
