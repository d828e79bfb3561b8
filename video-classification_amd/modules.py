"""Drop-in ``nn.Module`` classes for the reference's hot path, running on libvc_hip.

Class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys/shapes/dtypes
follow the reference (paths relative to its checkout):

  EncoderCNN / ResCNNEncoder / DecoderRNN / TCG_LSTM     utils/common/model_components.py:43,145,194,478
  Conv3DModel                                            models/Conv3D/UCF101_3DCNN_refactored.py:57
  CRNNModel / ResNetCRNNModel                            models/CRNN/UCF101_CRNN_refactored.py:64,
                                                         models/ResNetCRNN/UCF101_ResNetCRNN_refactored.py:63

The ``torch.nn`` sub-modules created here (Conv2d, BatchNorm2d, Linear, LSTM ...) are parameter
containers only -- they give the checkpoints their keys -- and their own ``forward`` is never
called: every layer is evaluated by the hand-written HIP kernels through ``ops``.  The encoders run
all B*T frames of a clip batch in one pass (frames are laid out time-major) instead of the
reference's per-time-step Python loop; BatchNorm keeps the loop's semantics by taking its
statistics per time step (``groups = T``) and applying the T running-statistic updates in order.

``compute_dtype`` is ``torch.bfloat16`` (MFMA path, fp32 accumulation, fp32 master weights) or
``torch.float32`` (exact path used by the parity tests).
"""
import torch
import torch.nn as nn

from . import ops

DEFAULT_COMPUTE_DTYPE = torch.bfloat16


def conv_out(size, k, s, p=0):
    return (size + 2 * p - (k - 1) - 1) // s + 1


class _HipModule(nn.Module):
    compute_dtype = DEFAULT_COMPUTE_DTYPE

    def set_compute_dtype(self, dtype):
        for m in self.modules():
            if isinstance(m, _HipModule):
                m.compute_dtype = dtype
        return self

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        ops.bump_weights_epoch()


def _frames_to_seq(y, T, B):
    """[T*B, E] time-major rows -> [B, T, E] (a view; the memory stays time-major)."""
    return y.view(T, B, y.shape[-1]).transpose(0, 1)


# --------------------------------------------------------------------------------------
# EncoderCNN
# --------------------------------------------------------------------------------------
class EncoderCNN(_HipModule):
    def __init__(self, img_x=224, img_y=224, fc_hidden1=512, fc_hidden2=512, drop_p=0.3, CNN_embed_dim=300,
                 use_pretrained=False):
        super().__init__()
        if use_pretrained:
            raise NotImplementedError("EncoderCNN(use_pretrained=True) needs torchvision's VGG16 weights "
                                      "(reference model_components.py:72-79); only the custom CNN is built here")
        self.img_x, self.img_y, self.CNN_embed_dim = img_x, img_y, CNN_embed_dim
        self.fc_hidden1, self.fc_hidden2, self.drop_p = fc_hidden1, fc_hidden2, drop_p
        self.features = None
        chans = (3, 32, 64, 128, 256)
        self.kernels = (5, 3, 3, 3)
        h, w = img_x, img_y
        for i, k in enumerate(self.kernels):
            setattr(self, f"conv{i + 1}", nn.Sequential(
                nn.Conv2d(chans[i], chans[i + 1], kernel_size=k, stride=2, padding=0),
                nn.BatchNorm2d(chans[i + 1], momentum=0.01), nn.ReLU(inplace=True)))
            h, w = conv_out(h, k, 2), conv_out(w, k, 2)
        self.feat_hw = (h, w)
        self.feature_dim = chans[-1] * h * w
        self.fc1 = nn.Linear(self.feature_dim, fc_hidden1)
        self.fc2 = nn.Linear(fc_hidden1, fc_hidden2)
        self.fc3 = nn.Linear(fc_hidden2, CNN_embed_dim)

    def forward(self, x_3d):
        B, T = x_3d.shape[0], x_3d.shape[1]
        a = None
        for i, k in enumerate(self.kernels):
            blk = getattr(self, f"conv{i + 1}")
            if i == 0:  # 3-channel layer: taps along W folded into the channel axis by the layout kernel
                a = ops.input_conv(x_3d, blk[0].weight, blk[0].bias, (1, k, k), (1, 2, 2), 0, "clip",
                                   self.compute_dtype, bn=blk[1], groups=T, relu=True)
            else:
                a = ops.conv_bn(a, blk[0].weight, blk[1], T, blk[0].bias, (1, k, k), (1, 2, 2), 0, relu=True)
        a = ops.mark_cut(a)  # fc1 below holds four fifths of the model's parameters: its gradient bucket completes here
        # flatten(NCHW) + fc1 == a convolution whose kernel covers the whole feature map
        hf, wf = self.feat_hw
        y = ops.conv(a, self.fc1.weight, self.fc1.bias, (1, hf, wf), 1, 0, relu=True, ci=a.shape[-1])
        y = y.view(T * B, y.shape[-1])
        y = ops.linear(y, self.fc2.weight, self.fc2.bias, relu=True)
        y = ops.dropout(y, self.drop_p, self.training)
        y = ops.linear(y, self.fc3.weight, self.fc3.bias, out_f32=True)[:, :self.CNN_embed_dim]
        return _frames_to_seq(y, T, B)


# --------------------------------------------------------------------------------------
# Bottleneck ResNet trunk (torchvision parameter names)
# --------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def run(self, x, T):
        s = self.stride
        # x feeds conv1 and the shortcut: both backward paths add into one gradient buffer (no add pass)
        join = ops.GradJoin(2, getattr(x, "_vc_bn_tail", None)) if (x.requires_grad and torch.is_grad_enabled()) else None
        idt = None
        if self.downsample is not None:
            # evaluated before conv1 so that its backward runs after conv1's: the strided 1x1 data gradient then
            # only ADDS into the pixels it reaches instead of also zero-filling the rest of the buffer
            idt = ops.conv_bn(x, self.downsample[0].weight, self.downsample[1], T, None, 1, (1, s, s), 0, join=join)
        # conv1 -> conv2 -> conv3 is a chain: each activation has one consumer, whose data-gradient kernel also
        # does the backward reductions of the BatchNorm in front of it
        a = ops.conv_bn(x, self.conv1.weight, self.bn1, T, relu=True, join=join, single_use=True)
        a = ops.conv_bn(a, self.conv2.weight, self.bn2, T, None, (1, 3, 3), (1, s, s), (0, 1, 1), relu=True,
                        single_use=True)
        # The block's output feeds the next block through a GradJoin, the average pool or a caller's own op, each of
        # which hands back a gradient tensor nobody else holds: bn3's backward may mask it in place.
        if idt is not None:
            return ops.conv_bn(a, self.conv3.weight, self.bn3, T, relu=True, residual=idt, grad_inplace=True)
        return ops.conv_bn(a, self.conv3.weight, self.bn3, T, relu=True, residual=x, res_join=join, grad_inplace=True)


RESNET_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def make_resnet_trunk(arch="resnet152", width=64):
    """Children 0..8 = conv1, bn1, relu, maxpool, layer1..4, avgpool, i.e. the layout of
    ``nn.Sequential(*list(resnet.children())[:-1])`` (reference model_components.py:164-165).  `width` = channels of
    the stem (64 in torchvision's networks; a multiple of 8)."""
    inplanes = width
    stages = []
    for i, (planes, n) in enumerate(zip((width, 2 * width, 4 * width, 8 * width), RESNET_LAYERS[arch])):
        blocks = []
        for b in range(n):
            s = (1 if i == 0 else 2) if b == 0 else 1
            ds = None
            if b == 0 and (s != 1 or inplanes != planes * 4):
                ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=s, bias=False), nn.BatchNorm2d(planes * 4))
            blocks.append(Bottleneck(inplanes, planes, s, ds))
            inplanes = planes * 4
        stages.append(nn.Sequential(*blocks))
    trunk = nn.Sequential(nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(width),
                          nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1), *stages,
                          nn.AdaptiveAvgPool2d((1, 1)))
    for m in trunk.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return trunk


def run_resnet_trunk(trunk, x_3d, T, dtype):
    """x_3d: raw fp32 clip [B,T,3,H,W] -> pooled features [T*B, 2048] (frames time-major)."""
    # conv1 -> bn1 -> relu -> maxpool(3, 2, 1): the last three are one pass (the 112 x 112 activation is never written)
    a = ops.input_conv(x_3d, trunk[0].weight, None, (1, 7, 7), (1, 2, 2), (0, 3, 3), "clip", dtype, bn=trunk[1],
                       groups=T, relu=True, pool=(3, 2, 1))
    for li in range(4, 8):
        for blk in trunk[li]:
            a = blk.run(a, T)
        if li < 7:  # stage boundary: the backward pass may be cut here (gradient buckets of the later stages complete)
            a = ops.mark_cut(a)
    F_, _, H, W, C = a.shape
    return ops.global_avg_pool(a.view(F_, H * W, C))


class _HeadBN(_HipModule):
    def _make_head(self, feat, h1, h2, embed):
        self.fc1 = nn.Linear(feat, h1)
        self.bn1 = nn.BatchNorm1d(h1, momentum=0.01)
        self.fc2 = nn.Linear(h1, h2)
        self.bn2 = nn.BatchNorm1d(h2, momentum=0.01)
        self.fc3 = nn.Linear(h2, embed)
        self._embed = embed

    def _head(self, f, T, drop_after_first):
        y = ops.batch_norm(ops.linear(f, self.fc1.weight, self.fc1.bias), self.bn1, T, relu=True)
        if drop_after_first:
            y = ops.dropout(y, self.drop_p, self.training)
        y = ops.batch_norm(ops.linear(y, self.fc2.weight, self.fc2.bias), self.bn2, T, relu=True)
        y = ops.dropout(y, self.drop_p, self.training)
        return ops.linear(y, self.fc3.weight, self.fc3.bias, out_f32=True)[:, :self._embed]


class ResCNNEncoder(_HeadBN):
    def __init__(self, fc_hidden1=512, fc_hidden2=512, drop_p=0.3, CNN_embed_dim=300, use_pretrained=True,
                 arch="resnet152", width=64):
        super().__init__()
        # The reference downloads ImageNet weights when use_pretrained is set (model_components.py:157-162);
        # here the trunk always starts from torchvision's random initialisation and pretrained weights, if
        # wanted, arrive through load_state_dict (same keys).
        self.fc_hidden1, self.fc_hidden2, self.drop_p = fc_hidden1, fc_hidden2, drop_p
        self.resnet = make_resnet_trunk(arch, width)
        self._make_head(32 * width, fc_hidden1, fc_hidden2, CNN_embed_dim)

    def forward(self, x_3d):
        B, T = x_3d.shape[0], x_3d.shape[1]
        f = run_resnet_trunk(self.resnet, x_3d, T, self.compute_dtype)
        return _frames_to_seq(self._head(f, T, False), T, B)


# --------------------------------------------------------------------------------------
# DecoderRNN
# --------------------------------------------------------------------------------------
class DecoderRNN(_HipModule):
    def __init__(self, CNN_embed_dim=300, h_RNN_layers=3, h_RNN=256, h_FC_dim=128, drop_p=0.3, num_classes=50):
        super().__init__()
        self.RNN_input_size, self.h_RNN_layers, self.h_RNN = CNN_embed_dim, h_RNN_layers, h_RNN
        self.h_FC_dim, self.drop_p, self.num_classes = h_FC_dim, drop_p, num_classes
        self.LSTM = nn.LSTM(input_size=CNN_embed_dim, hidden_size=h_RNN, num_layers=h_RNN_layers, batch_first=True)
        self.fc1 = nn.Linear(h_RNN, h_FC_dim)
        self.fc2 = nn.Linear(h_FC_dim, num_classes)

    def forward(self, x_RNN):
        x_tm = ops.cast_ag(x_RNN.transpose(0, 1).contiguous(), self.compute_dtype)   # [T,B,E]
        seq = ops.lstm(x_tm, self.LSTM)                                              # [T,B,H]
        y = ops.linear(seq[-1], self.fc1.weight, self.fc1.bias, relu=True)
        y = ops.dropout(y, self.drop_p, self.training)
        return ops.linear(y, self.fc2.weight, self.fc2.bias, out_f32=True)[:, :self.num_classes]


# --------------------------------------------------------------------------------------
# TCG-LSTM
# --------------------------------------------------------------------------------------
class TCG_LSTM(_HipModule):
    def __init__(self, embed_dim=512, hidden=512, layers=2, k_tokens=8, down=4, num_classes=4, drop=0.1):
        super().__init__()
        self.down, self.num_classes, self.drop = down, num_classes, drop
        self.lstm_short = nn.LSTM(2 * embed_dim, hidden, layers, batch_first=True, dropout=drop, bidirectional=True)
        self.lstm_long = nn.LSTM(embed_dim, hidden // 2, layers, batch_first=True, dropout=drop, bidirectional=True)
        self.gate = nn.Sequential(nn.Linear(4 * hidden, 2 * hidden), nn.Sigmoid())
        self.fc = nn.Sequential(nn.Linear(2 * hidden, 256), nn.ReLU(True), nn.Dropout(drop),
                                nn.Linear(256, num_classes))

    def forward(self, g, c):
        s_in, l_in = ops.tcg_inputs(g, c, self.down, self.compute_dtype)
        h_s = ops.lstm(s_in, self.lstm_short)          # [T,B,2H]
        h_l = ops.lstm(l_in, self.lstm_long)           # [Tl,B,H]
        clip = ops.tcg_fuse(h_s, h_l, self.gate[0])    # [B,2H]
        y = ops.linear(clip, self.fc[0].weight, self.fc[0].bias, relu=True)
        y = ops.dropout(y, self.drop, self.training)
        return ops.linear(y, self.fc[3].weight, self.fc[3].bias, out_f32=True)[:, :self.num_classes]


# --------------------------------------------------------------------------------------
# Conv3D
# --------------------------------------------------------------------------------------
class Conv3DModel(_HipModule):
    def __init__(self, config):
        super().__init__()
        mods, cin = [], 1
        for cout in (32, 64, 128, 256):
            mods += [nn.Conv3d(cin, cout, kernel_size=(3, 3, 3), padding=(1, 1, 1)), nn.BatchNorm3d(cout),
                     nn.ReLU(inplace=True), nn.MaxPool3d(kernel_size=(1, 2, 2))]
            cin = cout
        mods.append(nn.AdaptiveAvgPool3d((1, 1, 1)))
        self.conv3d = nn.Sequential(*mods)
        self.dropout_p, self.num_classes = config.dropout_p, config.num_classes
        self.fc = nn.Sequential(nn.Dropout(config.dropout_p), nn.Linear(256, 512), nn.ReLU(),
                                nn.Dropout(config.dropout_p), nn.Linear(512, config.num_classes))

    def forward(self, x):
        a = None
        for i in range(4):
            cv, bn = self.conv3d[4 * i], self.conv3d[4 * i + 1]
            # BatchNorm3d + ReLU + MaxPool3d((1,2,2)) as one pass: the pool never crosses the depth axis
            if i == 0:  # 1-channel layer straight from the fp32 volume (W taps folded into the channel axis)
                a = ops.input_conv(x, cv.weight, cv.bias, 3, 1, 1, "volume", self.compute_dtype, bn=bn, groups=1,
                                   relu=True, pool=(2, 2, 0))
            else:
                a = ops.conv_bn(a, cv.weight, bn, 1, cv.bias, 3, 1, 1, relu=True, pool=(2, 2, 0))
        B, D, H, W, C = a.shape
        f = ops.global_avg_pool(a.view(B, D * H * W, C))
        f = ops.dropout(f, self.dropout_p, self.training)
        y = ops.linear(f, self.fc[1].weight, self.fc[1].bias, relu=True)
        y = ops.dropout(y, self.dropout_p, self.training)
        return ops.linear(y, self.fc[4].weight, self.fc[4].bias, out_f32=True)[:, :self.num_classes]


# --------------------------------------------------------------------------------------
# encoder -> decoder wrappers
# --------------------------------------------------------------------------------------
class _EncDec(_HipModule):
    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder

    def forward(self, x):
        out = self.encoder(x)
        return self.decoder(*out) if isinstance(out, tuple) else self.decoder(out)


class CRNNModel(_EncDec):
    def __init__(self, config):
        super().__init__(
            EncoderCNN(img_x=config.img_x, img_y=config.img_y, fc_hidden1=config.CNN_fc_hidden1,
                       fc_hidden2=config.CNN_fc_hidden2, drop_p=config.dropout_p, CNN_embed_dim=config.CNN_embed_dim,
                       use_pretrained=getattr(config, "use_pretrained", False)),
            DecoderRNN(CNN_embed_dim=config.CNN_embed_dim, h_RNN_layers=config.RNN_hidden_layers,
                       h_RNN=config.RNN_hidden_nodes, h_FC_dim=config.RNN_FC_dim, drop_p=config.dropout_p,
                       num_classes=config.num_classes))


class ResNetCRNNModel(_EncDec):
    def __init__(self, config):
        super().__init__(
            ResCNNEncoder(fc_hidden1=config.CNN_fc_hidden1, fc_hidden2=config.CNN_fc_hidden2, drop_p=config.dropout_p,
                          CNN_embed_dim=config.CNN_embed_dim, use_pretrained=getattr(config, "use_pretrained", True),
                          arch=getattr(config, "arch", "resnet152")),
            DecoderRNN(CNN_embed_dim=config.CNN_embed_dim, h_RNN_layers=config.RNN_hidden_layers,
                       h_RNN=config.RNN_hidden_nodes, h_FC_dim=config.RNN_FC_dim, drop_p=config.dropout_p,
                       num_classes=config.num_classes))


class SwinTransformerRNNModel(_EncDec):
    """reference train.py:267-276"""


class SwinTCGModel(_EncDec):
    """reference train.py:313-321"""
