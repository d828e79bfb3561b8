"""The FLOP model behind the roofline fractions (video_classification_amd/flops.py) against figures counted
independently from the layer shapes: 2 x multiply-accumulates of convolutions, linear layers and LSTM gate
products, one clip = 16 frames."""
import types

import pytest

from video_classification_amd import flops as F
from video_classification_amd import modules as M
from video_classification_amd import swin as S


def test_encoder_cnn_and_decoder():
    enc = M.EncoderCNN(224, 224, 1024, 768, 0.0, 512)
    # conv 58.08 + 107.50 + 99.68 + 84.93, fc 75.50 + 1.57 + 0.79 MFLOP per frame
    assert F.encoder_cnn_flops(enc, 224, 224) / 1e6 == pytest.approx(428.05, abs=0.01)
    assert F.flops_per_clip(enc) / 1e9 == pytest.approx(6.849, abs=1e-3)
    assert F.flops_per_clip(M.DecoderRNN(512, 3, 512, 256, 0.0, 101)) / 1e9 == pytest.approx(0.2016, abs=2e-4)
    assert F.flops_per_clip(M.DecoderRNN(512, 2, 512, 256, 0.0, 101)) / 1e9 == pytest.approx(0.1345, abs=4e-4)


def test_tcg_lstm_and_conv3d():
    assert F.flops_per_clip(M.TCG_LSTM(96, 512, 2, 8, 4, 101, 0.1)) / 1e9 == pytest.approx(0.3796, abs=2e-4)
    assert F.flops_per_clip(M.TCG_LSTM(512, 512, 3, 8, 4, 101, 0.1)) / 1e9 == pytest.approx(0.7094, abs=2e-4)
    c3d = M.Conv3DModel(types.SimpleNamespace(dropout_p=0.5, num_classes=101))
    assert F.flops_per_clip(c3d, 16, 112, 112) / 1e9 == pytest.approx(16.99, abs=0.01)
    assert F.flops_per_clip(c3d, 16, 224, 224) / 1e9 == pytest.approx(67.98, abs=0.02)


def test_resnet_and_swin_trunks():
    # published multiply-accumulate counts of the architectures: ResNet-50 4.09 G, ResNet-152 11.5 G (without the
    # 1000-way classifier), Swin-T 4.5 G per 224 x 224 frame
    r50 = M.make_resnet_trunk("resnet50")
    assert F.resnet_trunk_flops(r50, 224, 224) / 2e9 == pytest.approx(4.09, abs=0.03)
    r152 = M.make_resnet_trunk("resnet152")
    assert F.resnet_trunk_flops(r152, 224, 224) / 2e9 == pytest.approx(11.5, abs=0.1)
    sw = S.SwinT()
    assert F.swin_flops(sw, 224, 224) / 2e9 == pytest.approx(4.5, abs=0.15)
