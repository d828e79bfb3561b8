"""CPU oracle: a plain-PyTorch fp32 restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``video_classification_amd`` imports this file; it is
used by ``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` as the checker / CPU baseline, never as the product path.

Every class restates one reference symbol (paths relative to the reference checkout):

  EncoderCNN              utils/common/model_components.py:43-142  (custom-CNN branch only)
  ResCNNEncoder           utils/common/model_components.py:145-191
  DecoderRNN              utils/common/model_components.py:194-231
  SwinTransformerEncoder  utils/common/model_components.py:234-309
  ClusterSwinEncoder      utils/common/model_components.py:408-475
  TCG_LSTM                utils/common/model_components.py:478-515
  Conv3DModel             models/Conv3D/UCF101_3DCNN_refactored.py:57-107
  CRNNModel               models/CRNN/UCF101_CRNN_refactored.py:64-94
  ResNetCRNNModel         models/ResNetCRNN/UCF101_ResNetCRNN_refactored.py:63-91
  SwinTransformerRNNModel train.py:267-276
  SwinTCGModel            train.py:313-321
  train_step / eval_step  utils/common/training_utils.py:71-92, 133-146

Pinning.  EncoderCNN, DecoderRNN, TCG_LSTM, Conv3DModel, the encoder FC/BN heads, CRNNModel and
the train step are pinned by golden vectors generated from the reference's own classes
(tests/golden/make_golden.py, run in the build container).  The ResNet trunk (torchvision
``resnet152``/``resnet50``) and the Swin-T backbone (timm ``swin_tiny_patch4_window7_224``) are
third-party code that is absent from the reference checkout and from this image (torchvision:
``requirements.txt`` pins only >=0.10.0; timm: not listed at all).  They are restated here from
the published architectures with torchvision / timm parameter names: PARITY UNPINNED for those
two sub-networks (self-consistency only).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def conv_out(size, k, s, p=0):
    """Output extent of a convolution (model_components.py:25-40)."""
    return (size + 2 * p - (k - 1) - 1) // s + 1


# --------------------------------------------------------------------------------------
# bf16 storage emulation (test infrastructure for the bf16 compute path of the HIP kernels)
# --------------------------------------------------------------------------------------
# With EMULATE_BF16 set, the restatement below keeps computing in fp32 but rounds to bfloat16 wherever the HIP path
# STORES a tensor in bf16: the clip, the packed weights, every convolution / linear / normalisation / pooling /
# attention output, the LSTM's h_t, and -- through the backward of the same rounding nodes -- the activation
# gradients at the same places (gate gradients of the LSTM included).  Accumulation, statistics, the LSTM cell
# state, logits and loss stay fp32, as on the device.  Off (the default), every helper is the identity and the
# oracle is the fp32 restatement the golden vectors pin.  The emulation is approximate where the device rounds a
# sum in two steps (gradients accumulated into a bf16 buffer by two kernels, a residual added to an already
# rounded GEMM result): there it rounds once.
EMULATE_BF16 = False


def _bf(x):
    return x.bfloat16().float()


class _Round(torch.autograd.Function):
    """value stored in bf16, and its gradient stored in bf16"""

    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


class _RoundValue(torch.autograd.Function):
    """bf16 copy of an fp32 master tensor (packed weights): the gradient stays fp32"""

    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """fp32 value whose gradient is stored in bf16 (LSTM gate pre-activations)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


def _r(x):
    return _Round.apply(x) if EMULATE_BF16 else x


def _w(w):
    return _RoundValue.apply(w) if (EMULATE_BF16 and w is not None) else w


def _rg(x):
    return _RoundGrad.apply(x) if EMULATE_BF16 else x


def _conv(m, x):
    """m(x) for an nn.Conv2d / nn.Conv3d, output stored in the compute type"""
    if not EMULATE_BF16:
        return m(x)
    f = F.conv3d if isinstance(m, nn.Conv3d) else F.conv2d
    return _r(f(x, _w(m.weight), m.bias, m.stride, m.padding))


def _lin(m, x, f32_out=False, relu=False):
    """m(x) for an nn.Linear (optionally followed by ReLU); f32_out: the device writes this output in fp32"""
    if not EMULATE_BF16:
        y = m(x)
        return F.relu(y) if relu else y
    y = F.linear(x, _w(m.weight), m.bias)
    if relu:
        y = F.relu(y)
    return y if f32_out else _r(y)


def _lstm(m, x):
    """nn.LSTM(batch_first) with zero initial state.  Emulation: the recurrence written out, with h_t stored in bf16
    (it is what the next step, the next layer and the output read), gate pre-activations and the cell state in fp32,
    gate gradients stored in bf16."""
    if not EMULATE_BF16:
        return m(x, None)[0]
    B, T, _ = x.shape
    H = m.hidden_size
    inp = _r(x)
    for l in range(m.num_layers):
        outs = []
        for d in range(2 if m.bidirectional else 1):
            sfx = f"_l{l}" + ("_reverse" if d else "")
            w_ih, w_hh = _w(getattr(m, "weight_ih" + sfx)), _w(getattr(m, "weight_hh" + sfx))
            bias = getattr(m, "bias_ih" + sfx) + getattr(m, "bias_hh" + sfx)
            gx = F.linear(inp, w_ih, bias)                       # [B,T,4H] fp32 on the device too
            h = x.new_zeros(B, H)
            c = x.new_zeros(B, H)
            hs = [None] * T
            for t in (range(T - 1, -1, -1) if d else range(T)):
                gates = _rg(gx[:, t] + F.linear(h, w_hh))
                i, f, g, o = gates.chunk(4, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = _r(torch.sigmoid(o) * torch.tanh(c))
                hs[t] = h
            outs.append(torch.stack(hs, dim=1))
        inp = torch.cat(outs, dim=-1) if len(outs) > 1 else outs[0]
        # inter-layer dropout is only ever exercised at p = 0 / eval in the parity tests
    return inp


def _per_frame(fn, clip):
    """Apply fn to every time step of [B,T,...] and restack as [B,T,E].

    The reference runs its encoders inside `for t in range(T)` (model_components.py:119,
    174, 281): BatchNorm statistics are per time step and running statistics receive T
    successive momentum updates per forward."""
    return torch.stack([fn(clip[:, t]) for t in range(clip.size(1))], dim=1)


# --------------------------------------------------------------------------------------
# EncoderCNN (custom CNN branch)
# --------------------------------------------------------------------------------------
class EncoderCNN(nn.Module):
    def __init__(self, img_x=224, img_y=224, fc_hidden1=512, fc_hidden2=512, drop_p=0.3, CNN_embed_dim=300,
                 use_pretrained=False):
        super().__init__()
        if use_pretrained:
            raise NotImplementedError("VGG16 branch needs torchvision weights (model_components.py:72-79)")
        self.drop_p = drop_p
        chans = (3, 32, 64, 128, 256)
        kernels = (5, 3, 3, 3)
        h, w = img_x, img_y
        for i, k in enumerate(kernels):
            block = nn.Sequential(nn.Conv2d(chans[i], chans[i + 1], k, stride=2, padding=0),
                                  nn.BatchNorm2d(chans[i + 1], momentum=0.01), nn.ReLU(inplace=True))
            setattr(self, f"conv{i + 1}", block)
            h, w = conv_out(h, k, 2), conv_out(w, k, 2)
        self.feature_dim = chans[-1] * h * w
        self.fc1 = nn.Linear(self.feature_dim, fc_hidden1)
        self.fc2 = nn.Linear(fc_hidden1, fc_hidden2)
        self.fc3 = nn.Linear(fc_hidden2, CNN_embed_dim)

    def _frame(self, x):
        x = _r(x)
        for blk in (self.conv1, self.conv2, self.conv3, self.conv4):  # Conv2d, BatchNorm2d, ReLU
            x = _r(blk[2](blk[1](_conv(blk[0], x))))
        x = x.flatten(1)
        x = _lin(self.fc2, _lin(self.fc1, x, relu=True), relu=True)
        return _lin(self.fc3, F.dropout(x, p=self.drop_p, training=self.training), f32_out=True)

    def forward(self, x_3d):
        return _per_frame(self._frame, x_3d)


# --------------------------------------------------------------------------------------
# Bottleneck ResNet trunk (torchvision naming; published architecture, parity unpinned)
# --------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)  # stride on 3x3 (v1.5)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else _r(self.downsample[1](_conv(self.downsample[0], x)))
        y = _r(self.relu(self.bn1(_conv(self.conv1, x))))
        y = _r(self.relu(self.bn2(_conv(self.conv2, y))))
        y = self.bn3(_conv(self.conv3, y))
        return _r(self.relu(y + idt))


RESNET_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def make_resnet_trunk(arch="resnet152", width=64):
    """`nn.Sequential(*list(resnet.children())[:-1])` (model_components.py:164-165): indices 0 conv1,
    1 bn1, 2 relu, 3 maxpool, 4-7 layer1-4, 8 avgpool.  `width` = channels of the stem (64 in torchvision; the
    stand-in trunks of the fixtures are narrower)."""
    layers = RESNET_LAYERS[arch]
    inplanes = width
    stages = []
    for i, (planes, n) in enumerate(zip((width, 2 * width, 4 * width, 8 * width), layers)):
        stride = 1 if i == 0 else 2
        blocks = []
        for b in range(n):
            ds = None
            s = stride if b == 0 else 1
            if b == 0 and (s != 1 or inplanes != planes * 4):
                ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=s, bias=False), nn.BatchNorm2d(planes * 4))
            blocks.append(Bottleneck(inplanes, planes, s, ds))
            inplanes = planes * 4
        stages.append(nn.Sequential(*blocks))
    trunk = nn.Sequential(nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(width),
                          nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1), *stages,
                          nn.AdaptiveAvgPool2d((1, 1)))
    for m in trunk.modules():  # torchvision's initialisation
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
    return trunk


class _HeadBN(nn.Module):
    """fc1 -> bn1 -> relu -> (dropout) -> fc2 -> bn2 -> relu -> dropout -> fc3 head shared by the ResNet and
    Swin encoders; `drop_after_first` distinguishes model_components.py:294-304 from :180-185."""

    def _head(self, x, drop_after_first):
        x = _r(F.relu(self.bn1(_lin(self.fc1, x))))
        if drop_after_first:
            x = F.dropout(x, p=self.drop_p, training=self.training)
        x = _r(F.relu(self.bn2(_lin(self.fc2, x))))
        x = F.dropout(x, p=self.drop_p, training=self.training)
        return _lin(self.fc3, x, f32_out=True)

    def _make_head(self, feat, h1, h2, embed):
        self.fc1 = nn.Linear(feat, h1)
        self.bn1 = nn.BatchNorm1d(h1, momentum=0.01)
        self.fc2 = nn.Linear(h1, h2)
        self.bn2 = nn.BatchNorm1d(h2, momentum=0.01)
        self.fc3 = nn.Linear(h2, embed)


class ResCNNEncoder(_HeadBN):
    def __init__(self, fc_hidden1=512, fc_hidden2=512, drop_p=0.3, CNN_embed_dim=300, use_pretrained=True,
                 arch="resnet152", width=64):
        super().__init__()
        # use_pretrained only selects the initial weights in the reference (model_components.py:157-162);
        # there is no network here, so the trunk always starts from torchvision's random init.
        self.drop_p = drop_p
        self.resnet = make_resnet_trunk(arch, width)
        self._make_head(32 * width, fc_hidden1, fc_hidden2, CNN_embed_dim)

    def _trunk(self, f):
        if not EMULATE_BF16:
            return self.resnet(f)
        t = self.resnet
        a = _r(t[3](t[2](t[1](_conv(t[0], _r(f))))))  # conv1, bn1, relu, maxpool: one stored tensor
        for li in range(4, 8):
            a = t[li](a)
        return _r(t[8](a))

    def forward(self, x_3d):
        return _per_frame(lambda f: self._head(self._trunk(f).flatten(1), False), x_3d)


# --------------------------------------------------------------------------------------
# DecoderRNN
# --------------------------------------------------------------------------------------
class DecoderRNN(nn.Module):
    def __init__(self, CNN_embed_dim=300, h_RNN_layers=3, h_RNN=256, h_FC_dim=128, drop_p=0.3, num_classes=50):
        super().__init__()
        self.drop_p = drop_p
        self.LSTM = nn.LSTM(input_size=CNN_embed_dim, hidden_size=h_RNN, num_layers=h_RNN_layers, batch_first=True)
        self.fc1 = nn.Linear(h_RNN, h_FC_dim)
        self.fc2 = nn.Linear(h_FC_dim, num_classes)

    def forward(self, x_RNN):
        seq = _lstm(self.LSTM, x_RNN)  # zero initial state
        x = _lin(self.fc1, seq[:, -1, :], relu=True)
        return _lin(self.fc2, F.dropout(x, p=self.drop_p, training=self.training), f32_out=True)


# --------------------------------------------------------------------------------------
# Swin-T backbone (timm naming; published architecture, parity unpinned)
# --------------------------------------------------------------------------------------
def _rel_pos_index(ws):
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)  # [ws*ws, ws*ws]


def _shift_mask(H, W, ws, shift):
    """Additive mask (0 / -100) per window for the cyclically shifted layout: [nW, ws*ws, ws*ws]."""
    img = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    win = img.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class WindowAttention(nn.Module):
    def __init__(self, dim, heads, ws):
        super().__init__()
        self.heads, self.ws = heads, ws
        self.scale = (dim // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", _rel_pos_index(ws), persistent=False)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, mask):  # x: [nWin_total, N, C]
        Bw, N, C = x.shape
        qkv = _lin(self.qkv, x).reshape(Bw, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1)
        attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = (attn.view(Bw // nW, nW, self.heads, N, N) + mask[None, :, None]).view(Bw, self.heads, N, N)
        attn = _r(attn.softmax(dim=-1))  # the probabilities enter the P.V product in the compute type
        # the projection's result joins the residual stream in the caller (one stored tensor: x + proj)
        return F.linear(_r((attn @ v).transpose(1, 2).reshape(Bw, N, C)), _w(self.proj.weight), self.proj.bias)


class SwinMlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        return F.linear(_r(F.gelu(_lin(self.fc1, x))), _w(self.fc2.weight), self.fc2.bias)


class SwinBlock(nn.Module):
    def __init__(self, dim, res, heads, ws, shift):
        super().__init__()
        self.res, self.ws = res, ws
        self.shift = 0 if res <= ws else shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, heads, ws)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = SwinMlp(dim)
        self.register_buffer("attn_mask", _shift_mask(res, res, ws, self.shift) if self.shift else None,
                             persistent=False)

    drop_path = 0.0  # set by SwinT (timm's linear schedule over the blocks)

    def _drop(self, y):
        """timm's DropPath: in training a sample's branch is zeroed with probability p, else scaled by 1 / (1 - p)."""
        p = self.drop_path
        if p == 0.0 or not self.training:
            return y
        keep = y.new_empty((y.shape[0],) + (1,) * (y.dim() - 1)).bernoulli_(1.0 - p)
        return _r(y * (keep / (1.0 - p)))

    def forward(self, x):  # x: [B, H, W, C]
        B, H, W, C = x.shape
        ws = self.ws
        y = _r(self.norm1(x))
        if self.shift:
            y = torch.roll(y, (-self.shift, -self.shift), (1, 2))
        y = y.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
        y = self.attn(y, self.attn_mask)
        y = y.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
        if self.shift:
            y = torch.roll(y, (self.shift, self.shift), (1, 2))
        x = _r(x + self._drop(y))
        return _r(x + self._drop(self.mlp(_r(self.norm2(x)))))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):  # [B,H,W,C] -> [B,H/2,W/2,2C]
        B, H, W, C = x.shape
        x = x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)
        return _lin(self.reduction, _r(self.norm(x)))


class SwinStage(nn.Module):
    def __init__(self, dim, res, depth, heads, ws, merge):
        super().__init__()
        self.downsample = PatchMerging(dim // 2) if merge else nn.Identity()
        self.blocks = nn.Sequential(*[SwinBlock(dim, res, heads, ws, 0 if i % 2 == 0 else ws // 2)
                                      for i in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, stride=patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return _r(self.norm(_conv(self.proj, _r(x)).permute(0, 2, 3, 1)))


class SwinT(nn.Module):
    """swin_tiny_patch4_window7_224 with num_classes=0: returns [B, 768] (LN then mean over tokens)."""

    def __init__(self, img=224, patch=4, dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), ws=7, drop_path_rate=0.1):
        # drop_path_rate: timm's SwinTransformer default; timm.create_model(name, pretrained=True, num_classes=0)
        # (reference model_components.py:254, 412) does not change it.  The parity tests pass 0.
        super().__init__()
        self.patch_embed = PatchEmbed(dim, patch)
        res = img // patch
        self.layers = nn.Sequential(*[
            SwinStage(dim * 2 ** i, res // 2 ** i, depths[i], heads[i], ws, merge=i > 0) for i in range(len(depths))])
        self.num_features = dim * 2 ** (len(depths) - 1)
        self.norm = nn.LayerNorm(self.num_features)
        blocks = [b for st in self.layers for b in st.blocks]
        for i, b in enumerate(blocks):
            b.drop_path = drop_path_rate * i / max(len(blocks) - 1, 1)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.layers(self.patch_embed(x))
        return _r(_r(self.norm(x)).mean(dim=(1, 2)))


class SwinTransformerEncoder(_HeadBN):
    def __init__(self, fc_hidden1=512, fc_hidden2=512, drop_p=0.3, CNN_embed_dim=300,
                 swin_model_name="swin_tiny_patch4_window7_224", use_pretrained=True, swin_kwargs=None):
        super().__init__()
        self.drop_p = drop_p
        self.swin = SwinT(**(swin_kwargs or {}))
        self._make_head(self.swin.num_features, fc_hidden1, fc_hidden2, CNN_embed_dim)

    def forward(self, x_3d):
        return _per_frame(lambda f: self._head(self.swin(f).flatten(1), True), x_3d)


class ClusterSwinEncoder(nn.Module):
    def __init__(self, backbone="swin_tiny_patch4_window7_224", embed_dim=512, k_tokens=8, swin_kwargs=None):
        super().__init__()
        self.k_tokens = k_tokens
        self.backbone = SwinT(**(swin_kwargs or {}))
        self.proj = nn.Linear(self.backbone.num_features, embed_dim)

    def forward(self, x):
        g = _per_frame(lambda f: _lin(self.proj, self.backbone(f), f32_out=True), x)  # [B,T,E]
        c = g.unsqueeze(2).repeat(1, 1, self.k_tokens, 1)                   # model_components.py:468
        return g, c


# --------------------------------------------------------------------------------------
# TCG-LSTM
# --------------------------------------------------------------------------------------
class TCG_LSTM(nn.Module):
    def __init__(self, embed_dim=512, hidden=512, layers=2, k_tokens=8, down=4, num_classes=4, drop=0.1):
        super().__init__()
        self.down = down
        self.lstm_short = nn.LSTM(2 * embed_dim, hidden, layers, batch_first=True, dropout=drop, bidirectional=True)
        self.lstm_long = nn.LSTM(embed_dim, hidden // 2, layers, batch_first=True, dropout=drop, bidirectional=True)
        self.gate = nn.Sequential(nn.Linear(4 * hidden, 2 * hidden), nn.Sigmoid())
        self.fc = nn.Sequential(nn.Linear(2 * hidden, 256), nn.ReLU(True), nn.Dropout(drop),
                                nn.Linear(256, num_classes))

    def forward(self, g, c):
        h_s = _lstm(self.lstm_short, torch.cat([g, c.mean(2)], dim=-1))      # [B,T,2H]
        h_l = _lstm(self.lstm_long, g[:, ::self.down, :])                    # [B,T/down,H]
        up = F.interpolate(h_l.transpose(1, 2), size=h_s.size(1), mode="linear", align_corners=False).transpose(1, 2)
        up2 = torch.cat([up, up], dim=-1)
        gate = self.gate(torch.cat([h_s, up2], dim=-1))
        h = gate * h_s + (1 - gate) * up2
        alpha = torch.softmax(h.mean(dim=-1), dim=1).unsqueeze(-1)
        return self.fc((h * alpha).sum(1))


# --------------------------------------------------------------------------------------
# Conv3D
# --------------------------------------------------------------------------------------
class Conv3DModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        mods = []
        cin = 1
        for cout in (32, 64, 128, 256):
            mods += [nn.Conv3d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm3d(cout), nn.ReLU(inplace=True),
                     nn.MaxPool3d(kernel_size=(1, 2, 2))]
            cin = cout
        mods.append(nn.AdaptiveAvgPool3d((1, 1, 1)))
        self.conv3d = nn.Sequential(*mods)
        self.fc = nn.Sequential(nn.Dropout(config.dropout_p), nn.Linear(256, 512), nn.ReLU(),
                                nn.Dropout(config.dropout_p), nn.Linear(512, config.num_classes))

    def forward(self, x):
        if not EMULATE_BF16:
            return self.fc(self.conv3d(x).flatten(1))
        a = _r(x)
        for i in range(4):  # Conv3d, BatchNorm3d, ReLU, MaxPool3d: the pooled tensor is what the device stores
            cv, bn, act, pool = self.conv3d[4 * i:4 * i + 4]
            a = _r(pool(act(bn(_conv(cv, a)))))
        f = _r(self.conv3d[16](a).flatten(1))
        f = _lin(self.fc[1], self.fc[0](f), relu=True)
        return _lin(self.fc[4], self.fc[3](f), f32_out=True)


# --------------------------------------------------------------------------------------
# Wrappers and the training step
# --------------------------------------------------------------------------------------
class _EncDec(nn.Module):
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
    pass


class SwinTCGModel(_EncDec):
    pass


# ----------------------------------------------------------------------------------------------
# Image / density fusion CNN (models/swintransformer_RNN/image_density_fusion.py)
# ----------------------------------------------------------------------------------------------
class ChannelAttention(nn.Module):
    """image_density_fusion.py:14-34: sigmoid(mlp(avgpool(x)) + mlp(maxpool(x))), mlp = 1x1 conv C -> C/r -> C."""

    def __init__(self, channels, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(channels, channels // reduction, 1, bias=False), nn.ReLU(inplace=True),
                                nn.Conv2d(channels // reduction, channels, 1, bias=False))

    def forward(self, x):  # [N, C, H, W] -> [N, C, 1, 1]
        # (adaptive_max_pool2d / max(dim) send the gradient to the first maximal element, as the reference's modules do)
        return torch.sigmoid(self.fc(x.mean((2, 3), keepdim=True)) + self.fc(F.adaptive_max_pool2d(x, 1)))


class SpatialAttention(nn.Module):
    """image_density_fusion.py:37-51: sigmoid(conv7x7([mean over C, max over C]))."""

    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)

    def forward(self, x):  # [N, C, H, W] -> [N, 1, H, W]
        return torch.sigmoid(self.conv(torch.cat([x.mean(1, keepdim=True), x.max(1, keepdim=True)[0]], 1)))


class LocalFeatureExtraction(nn.Module):
    """image_density_fusion.py:54-87."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(cout)
        self.channel_attention = ChannelAttention(cout)
        self.spatial_attention = SpatialAttention()

    def forward(self, x):
        y = _r(F.relu(self.bn1(_conv(self.conv1, x))))
        y = _r(self.bn2(_conv(self.conv2, y)))
        y = y * self.channel_attention(y)       # (the device applies both gates and the ReLU in one pass, fp32)
        return _r(F.relu(y * self.spatial_attention(y)))


class MultiScaleFeatureAggregation(nn.Module):
    """image_density_fusion.py:90-128: 1x1 / 3x3 / 5x5 / 7x7 branches of cout/4 channels, concatenated."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1x1 = nn.Conv2d(cin, cout // 4, 1)
        self.conv3x3 = nn.Conv2d(cin, cout // 4, 3, padding=1)
        self.conv5x5 = nn.Conv2d(cin, cout // 4, 5, padding=2)
        self.conv7x7 = nn.Conv2d(cin, cout // 4, 7, padding=3)
        self.bn = nn.BatchNorm2d(cout)
        self.fusion_conv = nn.Conv2d(cout, cout, 1)
        self.fusion_bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        y = torch.cat([_conv(m, x) for m in (self.conv1x1, self.conv3x3, self.conv5x5, self.conv7x7)], 1)
        y = _r(F.relu(self.bn(y)))
        return _r(F.relu(self.fusion_bn(_conv(self.fusion_conv, y))))


class ImageDensityFusion(nn.Module):
    """image_density_fusion.py:131-257: [image, density map] -> conv-bn-relu -> LocalFeatureExtraction ->
    MultiScaleFeatureAggregation -> 1x1 conv-bn-relu -> global average pool."""

    def __init__(self, input_channels=3, density_channels=1, feature_channels=64, output_channels=512):
        super().__init__()
        fc = feature_channels
        self.initial_conv = nn.Sequential(nn.Conv2d(input_channels + density_channels, fc, 3, padding=1),
                                          nn.BatchNorm2d(fc), nn.ReLU(inplace=True))
        self.local_feature_extraction = LocalFeatureExtraction(fc, fc)
        self.multi_scale_aggregation = MultiScaleFeatureAggregation(fc, 2 * fc)
        self.final_conv = nn.Sequential(nn.Conv2d(2 * fc, output_channels, 1), nn.BatchNorm2d(output_channels),
                                        nn.ReLU(inplace=True), nn.AdaptiveAvgPool2d((1, 1)))
        self.output_dim = output_channels

    def forward(self, original_image, density_channel):  # [N,C,H,W], [N,1,h,w] -> [N, output_channels]
        if original_image.shape[-2:] != density_channel.shape[-2:]:
            density_channel = F.interpolate(density_channel, size=original_image.shape[-2:], mode="bilinear",
                                            align_corners=False)
        x = _r(torch.cat([original_image, density_channel], 1))
        x = _r(F.relu(self.initial_conv[1](_conv(self.initial_conv[0], x))))
        x = self.multi_scale_aggregation(self.local_feature_extraction(x))
        x = _r(F.relu(self.final_conv[1](_conv(self.final_conv[0], x))))
        return x.mean((2, 3))

    def forward_batch(self, video_frames, density_channels):  # [B,T,C,H,W], [B,T,1,h,w] -> [B, T, output_channels]
        B, T = video_frames.shape[:2]
        return self.forward(video_frames.flatten(0, 1), density_channels.flatten(0, 1)).view(B, T, -1)


def train_step(model, data, target, criterion, optimizer):
    """One batch of train_epoch (training_utils.py:73-92): returns (loss, logits, n_correct)."""
    model.train()
    optimizer.zero_grad()
    output = model(data)
    loss = criterion(output, target.squeeze())
    loss.backward()
    optimizer.step()
    pred = output.argmax(dim=1, keepdim=True)
    return loss.item(), output.detach(), pred.eq(target.view_as(pred)).sum().item()


@torch.no_grad()
def eval_step(model, data, target, criterion):
    """One batch of validate_epoch (training_utils.py:133-146)."""
    model.eval()
    output = model(data)
    loss = criterion(output, target.squeeze())
    pred = output.argmax(dim=1, keepdim=True)
    return loss.item(), output, pred.eq(target.view_as(pred)).sum().item()
