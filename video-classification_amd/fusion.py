"""``ImageDensityFusion`` of the reference (models/swintransformer_RNN/image_density_fusion.py:131-257) on libvc_hip.

[frame, density map] -> conv-bn-relu -> LocalFeatureExtraction (two 3x3 conv-bn, channel + spatial attention gate)
-> MultiScaleFeatureAggregation (1x1 / 3x3 / 5x5 / 7x7 branches, bn, 1x1 conv-bn) -> 1x1 conv-bn-relu -> global average
pool.  Same class names, constructor arguments, ``forward`` / ``forward_batch`` signatures and ``state_dict`` keys as
the reference; the ``torch.nn`` sub-modules only hold the parameters.  The convolutions and BatchNorms run on the
implicit-GEMM / BatchNorm kernels of ``ops``; the attention gate is ``vc_cbam_fwd`` / ``vc_cbam_bwd``.  The reference's
density / velocity proxy generators (OpenCV, CPU) are not part of this module: it takes the density map as an input,
as ``forward`` does.  BatchNorm statistics are taken over the whole call (``forward_batch`` flattens B*T, reference
:248-252).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .modules import _HipModule


class _Gate(torch.autograd.Function):
    """relu(x * channel_gate(x) * spatial_gate(x * channel_gate(x))) on channels-last x [N, 1, H, W, C]."""

    @staticmethod
    def forward(ctx, x, w1, w2, ws):
        ops._require_cuda(x, "attention gate")
        x = x.contiguous()
        N, _, H, W, C = x.shape
        ch, k = w1.shape[0], ws.shape[-1]
        dev, f32 = x.device, torch.float32
        P = H * W
        sv = dict(avg=torch.empty(N, C, device=dev, dtype=f32), mx=torch.empty(N, C, device=dev, dtype=f32),
                  argp=torch.empty(N, C, device=dev, dtype=torch.int32),
                  hid=torch.empty(N, 2, ch, device=dev, dtype=f32), gc=torch.empty(N, C, device=dev, dtype=f32),
                  pooled=torch.empty(N, P, 2, device=dev, dtype=f32),
                  argc=torch.empty(N, P, device=dev, dtype=torch.int32), gs=torch.empty(N, P, device=dev, dtype=f32))
        work = torch.empty(L.query("vc_cbam_workspace_floats", N, H, W, C), device=dev, dtype=f32)
        out = torch.empty_like(x)
        w1c, w2c, wsc = w1.detach().contiguous(), w2.detach().contiguous(), ws.detach().contiguous()
        L.call("vc_cbam_fwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(w1c), L.ptr(w2c), L.ptr(wsc), L.ptr(out),
               L.ptr(sv["avg"]), L.ptr(sv["mx"]), L.ptr(sv["argp"]), L.ptr(sv["hid"]), L.ptr(sv["gc"]),
               L.ptr(sv["pooled"]), L.ptr(sv["argc"]), L.ptr(sv["gs"]), L.ptr(work), N, H, W, C, ch, k, L.stream())
        ctx.dims = (N, H, W, C, ch, k)
        ctx.params = (w1, w2, ws)
        ctx.save_for_backward(x, w1c, w2c, wsc, *sv.values())
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1c, w2c, wsc, avg, mx, argp, hid, gc, pooled, argc, gs = ctx.saved_tensors
        N, H, W, C, ch, k = ctx.dims
        w1, w2, ws = ctx.params
        dev, f32 = x.device, torch.float32
        dout = ops.cast(dout.contiguous(), x.dtype)
        dx = torch.empty_like(x)
        P = H * W
        tmp = [torch.empty(N, P, device=dev, dtype=f32), torch.empty(N, P, 2, device=dev, dtype=f32),
               torch.empty(N, C, device=dev, dtype=f32), torch.empty(N, C, device=dev, dtype=f32),
               torch.empty(N, C, device=dev, dtype=f32)]
        L.call("vc_cbam_bwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(dout), L.ptr(w1c), L.ptr(w2c), L.ptr(wsc),
               L.ptr(avg), L.ptr(mx), L.ptr(argp), L.ptr(hid), L.ptr(gc), L.ptr(pooled), L.ptr(argc), L.ptr(gs),
               L.ptr(dx), L.ptr(ops._grad_buf(w1)), L.ptr(ops._grad_buf(w2)), L.ptr(ops._grad_buf(ws)),
               *[L.ptr(t) for t in tmp], N, H, W, C, ch, k, L.stream())
        ops._ready(w1, w2, ws)
        return dx, None, None, None


def attention_gate(x, w1, w2, ws):
    """x: [N,1,H,W,C] channels-last; w1 [C/r, C, 1, 1], w2 [C, C/r, 1, 1] (ChannelAttention.fc.0 / fc.2), ws [1, 2, k, k]
    (SpatialAttention.conv): out = relu(y * sa(y)), y = x * ca(x) (reference :81-86)."""
    return _Gate.apply(x, w1, w2, ws)


class ChannelAttention(nn.Module):  # parameter container (reference :14-34)
    def __init__(self, in_channels, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(in_channels, in_channels // reduction, 1, bias=False), nn.ReLU(inplace=True),
                                nn.Conv2d(in_channels // reduction, in_channels, 1, bias=False))


class SpatialAttention(nn.Module):  # parameter container (reference :37-51)
    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)


class LocalFeatureExtraction(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.channel_attention = ChannelAttention(out_channels)
        self.spatial_attention = SpatialAttention()

    def run(self, x):
        a = ops.conv_bn(x, self.conv1.weight, self.bn1, 1, self.conv1.bias, (1, 3, 3), 1, (0, 1, 1), relu=True,
                        single_use=True)
        y = ops.conv_bn(a, self.conv2.weight, self.bn2, 1, self.conv2.bias, (1, 3, 3), 1, (0, 1, 1))
        return attention_gate(y, self.channel_attention.fc[0].weight, self.channel_attention.fc[2].weight,
                              self.spatial_attention.conv.weight)


class MultiScaleFeatureAggregation(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        if (out_channels // 4) % 8:
            raise ValueError("MultiScaleFeatureAggregation: out_channels / 4 must be a multiple of 8 on this path")
        self.conv1x1 = nn.Conv2d(in_channels, out_channels // 4, 1)
        self.conv3x3 = nn.Conv2d(in_channels, out_channels // 4, 3, padding=1)
        self.conv5x5 = nn.Conv2d(in_channels, out_channels // 4, 5, padding=2)
        self.conv7x7 = nn.Conv2d(in_channels, out_channels // 4, 7, padding=3)
        self.bn = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.fusion_conv = nn.Conv2d(out_channels, out_channels, 1)
        self.fusion_bn = nn.BatchNorm2d(out_channels)
        self.fusion_relu = nn.ReLU(inplace=True)

    def run(self, x):
        # x feeds four convolutions: their data gradients accumulate in one buffer inside the kernels
        join = ops.GradJoin(4) if (x.requires_grad and torch.is_grad_enabled()) else None
        ys = [ops.conv(x, m.weight, m.bias, (1, k, k), 1, (0, k // 2, k // 2), join=join)
              for k, m in ((1, self.conv1x1), (3, self.conv3x3), (5, self.conv5x5), (7, self.conv7x7))]
        y = ops.batch_norm(torch.cat(ys, dim=-1), self.bn, 1, relu=True, single_use=True)
        return ops.conv_bn(y, self.fusion_conv.weight, self.fusion_bn, 1, self.fusion_conv.bias, relu=True,
                           single_use=True)


class ImageDensityFusion(_HipModule):
    def __init__(self, input_channels=3, density_channels=1, feature_channels=64, output_channels=512):
        super().__init__()
        self.input_channels, self.density_channels = input_channels, density_channels
        self.feature_channels, self.output_channels = feature_channels, output_channels
        if feature_channels % 16 or feature_channels < 16:
            raise ValueError("ImageDensityFusion: feature_channels must be a multiple of 16 (channel attention C / 16)")
        total = input_channels + density_channels
        self.initial_conv = nn.Sequential(nn.Conv2d(total, feature_channels, 3, padding=1),
                                          nn.BatchNorm2d(feature_channels), nn.ReLU(inplace=True))
        self.local_feature_extraction = LocalFeatureExtraction(feature_channels, feature_channels)
        self.multi_scale_aggregation = MultiScaleFeatureAggregation(feature_channels, feature_channels * 2)
        self.final_conv = nn.Sequential(nn.Conv2d(feature_channels * 2, output_channels, 1),
                                        nn.BatchNorm2d(output_channels), nn.ReLU(inplace=True),
                                        nn.AdaptiveAvgPool2d((1, 1)))
        self.output_dim = output_channels

    def forward(self, original_image, density_channel):
        """original_image [N, C, H, W], density_channel [N, 1, h, w] (fp32) -> fused features [N, output_channels] fp32."""
        ops._require_cuda(original_image, "ImageDensityFusion")
        if original_image.shape[-2:] != density_channel.shape[-2:]:
            # the density map's own resolution (32 x 32 in the reference's configuration) -> the frame's; an input
            # re-sampling step, no gradient flows into it
            density_channel = F.interpolate(density_channel, size=original_image.shape[-2:], mode="bilinear",
                                            align_corners=False)
        x = torch.cat([original_image, density_channel], dim=1).unsqueeze(2)        # [N, C + 1, 1, H, W]
        a = ops.volume_to_cl(x, self.compute_dtype, ops.pad8(x.shape[1]))            # [N, 1, H, W, 8]
        ic, fc = self.initial_conv, self.final_conv
        a = ops.conv_bn(a, ic[0].weight, ic[1], 1, ic[0].bias, (1, 3, 3), 1, (0, 1, 1), relu=True, single_use=True)
        a = self.local_feature_extraction.run(a)
        a = self.multi_scale_aggregation.run(a)
        a = ops.conv_bn(a, fc[0].weight, fc[1], 1, fc[0].bias, relu=True)
        N, _, H, W, C = a.shape
        f = ops.global_avg_pool(a.view(N, H * W, C))
        return ops.cast_ag(f, torch.float32)[:, :self.output_channels]

    def forward_batch(self, video_frames, density_channels):
        """video_frames [B, T, C, H, W], density_channels [B, T, 1, h, w] -> [B, T, output_channels]."""
        B, T = video_frames.shape[:2]
        f = self.forward(video_frames.reshape(-1, *video_frames.shape[2:]),
                         density_channels.reshape(-1, *density_channels.shape[2:]))
        return f.view(B, T, -1)


def create_image_density_fusion(input_channels=3, density_channels=1, feature_channels=64, output_channels=512):
    """reference :260-281"""
    return ImageDensityFusion(input_channels, density_channels, feature_channels, output_channels)


def combined_features(swin_features, fused_features=None, velocity_channels=None):
    """The feature list of the reference's train / validate loops (UCF101_SwinCRNN_fixed.py:232-272, :313-350):
    Swin features [B, T, E], optionally the fusion CNN's [B, T, F] and the velocity maps [B, T-1, 1, h, w] (last map
    repeated to T, flattened to [B, T, h*w]), concatenated along the feature axis for the decoder."""
    feats = [swin_features]
    if fused_features is not None:
        feats.append(fused_features.to(swin_features.dtype))
    if velocity_channels is not None:
        v = velocity_channels
        if v.shape[1] < swin_features.shape[1]:
            v = torch.cat([v, v[:, -1:]], dim=1)
        feats.append(v.reshape(v.shape[0], v.shape[1], -1).to(swin_features.dtype))
    return feats[0] if len(feats) == 1 else torch.cat(feats, dim=-1)


def forward_with_proxies(swin_encoder, rnn_decoder, X, density_fusion_module=None, density_channels=None,
                         velocity_channels=None):
    """One forward of the reference's loop body (:232-272) with the proxy maps supplied by the caller (the
    reference computes them with OpenCV on the host, density_proxy_generation.py / velocity_proxy_generation.py).
    The decoder's input width must equal the concatenated width, as in the reference."""
    fused = None
    if density_fusion_module is not None and density_channels is not None:
        fused = density_fusion_module.forward_batch(X, density_channels)
    return rnn_decoder(combined_features(swin_encoder(X), fused, velocity_channels))
