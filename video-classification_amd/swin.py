"""Swin-T backbone and the two Swin encoders of the reference on the HIP kernels.

  SwinTransformerEncoder   utils/common/model_components.py:234-309
  ClusterSwinEncoder       utils/common/model_components.py:408-475

The reference obtains the backbone from ``timm.create_model('swin_tiny_patch4_window7_224', pretrained=True,
num_classes=0)``; timm is not part of the reference checkout, so the architecture is restated here from the
published definition with timm's parameter names (patch_embed.proj/norm, layers.N.downsample.norm/reduction,
layers.N.blocks.M.norm1/attn.qkv/attn.proj/attn.relative_position_bias_table/norm2/mlp.fc1/mlp.fc2, norm), so
that a timm checkpoint loads by key.  Frames are batched over B*T (LayerNorm has no batch statistics, so this
is exact); the encoder heads' BatchNorm1d keeps per-time-step statistics.
"""
import torch
import torch.nn as nn

from . import ops
from .modules import _HeadBN, _HipModule, _frames_to_seq


class _Attn(nn.Module):
    def __init__(self, dim, heads, ws):
        super().__init__()
        self.heads, self.ws = heads, ws
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class SwinBlock(nn.Module):
    def __init__(self, dim, res, heads, ws, shift):
        super().__init__()
        self.res, self.ws = res, ws
        self.shift = 0 if res <= ws else shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attn(dim, heads, ws)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim)

    def run(self, x, drop_path, training):
        """x: [F, H, W, C] tokens in the compute dtype."""
        F_, H, W, C = x.shape
        t, y = ops.layer_norm_fork(x.view(F_ * H * W, C), self.norm1)
        qkv = ops.linear(y, self.attn.qkv.weight, self.attn.qkv.bias)
        a = ops.window_attention(qkv, self.attn.relative_position_bias_table, F_, H, W, self.attn.heads, self.ws,
                                 self.shift)
        # x + drop_path(proj(attn)) in the projection GEMM's epilogue: stochastic depth scales a frame's branch by one
        # factor (0 or 1 / keep) before it joins the residual stream
        dp = drop_path if training else 0.0
        t = ops.linear(a, self.attn.proj.weight, self.attn.proj.bias, residual=t, drop_path=dp, samples=F_)
        t, y = ops.layer_norm_fork(t, self.norm2)
        h, gh = ops.linear(y, self.mlp.fc1.weight, self.mlp.fc1.bias, want_gelu=True)  # fc1 and gelu(fc1)
        # fc2 is applied to gelu(h); the gradient through the GELU is taken in fc2's data-gradient GEMM
        t = ops.linear(h, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=t, pre_gelu=True, gelu_x=gh, drop_path=dp,
                       samples=F_)
        return t.view(F_, H, W, C)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def run(self, x):
        m = ops.patch_merge(x)                       # [F, H/2, W/2, 4C]
        F_, H2, W2, C4 = m.shape
        y = ops.layer_norm(m.view(F_ * H2 * W2, C4), self.norm)
        y = ops.linear(y, self.reduction.weight, None)
        return y.view(F_, H2, W2, y.shape[-1])


class SwinStage(nn.Module):
    def __init__(self, dim, res, depth, heads, ws, merge):
        super().__init__()
        self.downsample = PatchMerging(dim // 2) if merge else nn.Identity()
        self.blocks = nn.Sequential(*[SwinBlock(dim, res, heads, ws, 0 if i % 2 == 0 else ws // 2)
                                      for i in range(depth)])


class PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.patch = patch
        self.proj = nn.Conv2d(3, dim, patch, stride=patch)
        self.norm = nn.LayerNorm(dim)


class SwinT(nn.Module):
    """swin_tiny_patch4_window7_224 with num_classes=0: frames -> [F, 768] (final LayerNorm, mean over tokens)."""

    def __init__(self, img=224, patch=4, dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), ws=7,
                 drop_path_rate=0.1):
        # drop_path_rate: timm's SwinTransformer default, which timm.create_model(name, pretrained, num_classes=0)
        # (model_components.py:254, 412) leaves in place: the reference trains with stochastic depth 0 .. 0.1
        super().__init__()
        self.patch_embed = PatchEmbed(dim, patch)
        res = img // patch
        self.layers = nn.Sequential(*[
            SwinStage(dim * 2 ** i, res // 2 ** i, depths[i], heads[i], ws, merge=i > 0) for i in range(len(depths))])
        self.num_features = dim * 2 ** (len(depths) - 1)
        self.norm = nn.LayerNorm(self.num_features)
        n = sum(depths)
        self.drop_path = [drop_path_rate * i / max(n - 1, 1) for i in range(n)]  # timm's linear schedule
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """Accepts both timm checkpoint layouts of swin_tiny_patch4_window7_224 (the reference builds its backbone with
        an unpinned timm, model_components.py:254, 412): timm >= 0.9 puts a stage's PatchMerging at the START of
        layers.{1,2,3} and keeps no index / mask buffers; older timm puts it at the END of layers.{0,1,2} and stores
        persistent ``attn.relative_position_index`` and ``attn_mask`` buffers.  The buffers are derived from the
        geometry by the kernels, a classifier ``head`` does not exist at num_classes=0: those keys are dropped, the
        older downsample keys are renamed.  Runs before the children load, on the dict load_state_dict passes down."""
        mine = [k for k in state_dict if k.startswith(prefix)]
        for k in mine:
            rest = k[len(prefix):]
            if rest.endswith("attn.relative_position_index") or rest.endswith(".attn_mask") or rest.startswith("head."):
                del state_dict[k]
        if any(k.startswith(prefix + "layers.0.downsample.") for k in state_dict):  # the older layout
            for i in range(len(self.layers) - 2, -1, -1):
                src = f"{prefix}layers.{i}.downsample."
                for k in [k for k in state_dict if k.startswith(src)]:
                    state_dict[f"{prefix}layers.{i + 1}.downsample.{k[len(src):]}"] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def run(self, a):
        """a: channels-last frames [F,1,H,W,8] -> [F, num_features]."""
        pe = self.patch_embed
        p = pe.patch
        y = ops.conv(a, pe.proj.weight, pe.proj.bias, (1, p, p), (1, p, p), 0)      # [F,1,H/p,W/p,dim]
        F_, _, H, W, C = y.shape
        x = ops.layer_norm(y.view(F_ * H * W, C), pe.norm).view(F_, H, W, C)
        k = 0
        for si, stage in enumerate(self.layers):
            if si > 0:  # stage boundary: the backward pass may be cut here (gradient buckets of the later stages complete)
                x = ops.mark_cut(x)
            if not isinstance(stage.downsample, nn.Identity):
                x = stage.downsample.run(x)
            for blk in stage.blocks:
                x = blk.run(x, self.drop_path[k], self.training)
                k += 1
        F_, H, W, C = x.shape
        x = ops.layer_norm(x.view(F_ * H * W, C), self.norm)
        return ops.global_avg_pool(x.view(F_, H * W, C))


class SwinTransformerEncoder(_HeadBN):
    def __init__(self, fc_hidden1=512, fc_hidden2=512, drop_p=0.3, CNN_embed_dim=300,
                 swin_model_name="swin_tiny_patch4_window7_224", use_pretrained=True, swin_kwargs=None):
        super().__init__()
        if swin_model_name != "swin_tiny_patch4_window7_224":
            raise NotImplementedError(f"only swin_tiny_patch4_window7_224 is built (got {swin_model_name})")
        # pretrained ImageNet weights, if wanted, arrive through load_state_dict (timm keys under `swin.`)
        self.fc_hidden1, self.fc_hidden2, self.drop_p = fc_hidden1, fc_hidden2, drop_p
        self.swin = SwinT(**(swin_kwargs or {}))
        self.feature_dim = self.swin.num_features
        self._make_head(self.feature_dim, fc_hidden1, fc_hidden2, CNN_embed_dim)

    def forward(self, x_3d):
        B, T = x_3d.shape[0], x_3d.shape[1]
        f = self.swin.run(ops.clip_to_frames(x_3d, self.compute_dtype))
        return _frames_to_seq(self._head(f, T, True), T, B)


class ClusterSwinEncoder(_HipModule):
    def __init__(self, backbone="swin_tiny_patch4_window7_224", embed_dim=512, k_tokens=8, swin_kwargs=None):
        super().__init__()
        if backbone != "swin_tiny_patch4_window7_224":
            raise NotImplementedError(f"only swin_tiny_patch4_window7_224 is built (got {backbone})")
        self.embed_dim, self.k_tokens = embed_dim, k_tokens
        self.backbone = SwinT(**(swin_kwargs or {}))
        self.feature_dim = self.backbone.num_features
        self.proj = nn.Linear(self.feature_dim, embed_dim)

    def forward(self, x):
        B, T = x.shape[0], x.shape[1]
        f = self.backbone.run(ops.clip_to_frames(x, self.compute_dtype))
        g = _frames_to_seq(ops.linear(f, self.proj.weight, self.proj.bias, out_f32=True)[:, :self.embed_dim], T, B)
        # the reference materialises k identical copies (model_components.py:468); an expanded view has the
        # same values and shape [B,T,k,E] without the 8x copy (TCG_LSTM only takes their mean)
        c = g.unsqueeze(2).expand(B, T, self.k_tokens, self.embed_dim)
        return g, c
