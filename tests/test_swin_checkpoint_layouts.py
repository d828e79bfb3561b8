"""Swin-T checkpoints written under either timm layout load with strict=True (CPU; modules only, no kernels).

The reference takes its backbone from an unpinned timm (utils/common/model_components.py:254, 412): checkpoints of
timm >= 0.9 carry ``layers.{1,2,3}.downsample`` and no index / mask buffers, older ones ``layers.{0,1,2}.downsample``
plus persistent ``attn.relative_position_index`` / ``attn_mask`` buffers, and either may carry a ``head``.  Both
layouts are built here from the model's own state_dict (timm is not installed)."""
import re

import pytest
import torch

from video_classification_amd.swin import ClusterSwinEncoder, SwinT, SwinTransformerEncoder

KW = dict(img=56, patch=4, dim=16, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8), ws=7)


def old_layout(sd, prefix, heads=KW["heads"], ws=KW["ws"], res0=KW["img"] // KW["patch"]):
    """The pre-0.9 rendering of a state_dict whose backbone keys start with `prefix`."""
    out = {}
    for k, v in sd.items():
        m = re.match(re.escape(prefix) + r"layers\.(\d)\.downsample\.(.*)", k)
        out[f"{prefix}layers.{int(m.group(1)) - 1}.downsample.{m.group(2)}" if m else k] = v
    for i, h in enumerate(heads):
        for j in range(2):
            out[f"{prefix}layers.{i}.blocks.{j}.attn.relative_position_index"] = torch.zeros(ws * ws, ws * ws, dtype=torch.long)
            if j % 2 == 1 and res0 // 2 ** i > ws:
                n = (res0 // 2 ** i // ws) ** 2
                out[f"{prefix}layers.{i}.blocks.{j}.attn_mask"] = torch.zeros(n, ws * ws, ws * ws)
    out[prefix + "head.weight"] = torch.zeros(10, 16 * 8)
    out[prefix + "head.bias"] = torch.zeros(10)
    return out


def new_layout_with_extras(sd, prefix):
    out = dict(sd)
    out[prefix + "head.fc.weight"] = torch.zeros(10, 16 * 8)
    out[prefix + "head.fc.bias"] = torch.zeros(10)
    out[prefix + "layers.1.blocks.1.attn_mask"] = torch.zeros(4, 49, 49)
    out[prefix + "layers.0.blocks.0.attn.relative_position_index"] = torch.zeros(49, 49, dtype=torch.long)
    return out


def build(kind):
    if kind == "backbone":
        return SwinT(**KW), ""
    if kind == "swin_encoder":
        return SwinTransformerEncoder(32, 24, 0.0, 16, swin_kwargs=KW), "swin."
    return ClusterSwinEncoder(embed_dim=24, swin_kwargs=KW), "backbone."


@pytest.mark.parametrize("kind", ["backbone", "swin_encoder", "cluster_encoder"])
@pytest.mark.parametrize("layout", ["old", "new_with_extras", "own"])
def test_both_timm_layouts_load_strictly(kind, layout):
    torch.manual_seed(0)
    src, prefix = build(kind)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    ckpt = {"old": old_layout, "new_with_extras": new_layout_with_extras, "own": lambda s, p: dict(s)}[layout](sd, prefix)
    if layout == "old":
        assert any(k.startswith(prefix + "layers.0.downsample.") for k in ckpt)
        assert not any(k.startswith(prefix + "layers.3.downsample.") for k in ckpt)
    torch.manual_seed(1)
    dst, _ = build(kind)
    result = dst.load_state_dict(ckpt, strict=True)
    assert not result.missing_keys and not result.unexpected_keys
    got = dst.state_dict()
    assert set(got) == set(sd)  # our own keys stay the timm >= 0.9 ones
    for k in sd:
        assert torch.equal(got[k], sd[k]), k


def test_a_really_unknown_key_is_still_refused():
    src, _ = build("backbone")
    sd = dict(src.state_dict())
    sd["layers.0.blocks.0.attn.something_else"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="something_else"):
        build("backbone")[0].load_state_dict(sd, strict=True)
