"""The 28-frame shapes of the reference's other callers, against the CPU oracle with identical weights and inputs.

predict.py:55 and the stand-alone scripts' begin / end frame take clips of T = 28 frames (train.py uses 16): 28 BatchNorm
groups per encoder call, 28 recurrent steps, and the TCG-LSTM's long branch runs T / 4 = 7 steps whose outputs are
interpolated 7 -> 28 (model_components.py:496-503).  The stand-alone clusterSwin script builds
TCG_LSTM(embed_dim=512, hidden=512, layers=3) (models/clusterSwin_TCGlstm/clusterSwin_TCGlstm.py:228-236).

Tolerances as in tests/test_gpu_models.py: fp32 path 2e-4 outputs / 1e-3 gradients, bf16 path 4e-2 / 2e-1."""
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu

TOL = {torch.float32: (2e-4, 1e-3), torch.bfloat16: (4e-2, 2e-1)}
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]


def _fwd_bwd(ref, mine, inputs, cuda, dt, seed=5, cancelled=(), input_grads=True):
    tol_o, tol_g = TOL[dt]
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    ins_r = [x.clone().requires_grad_(x.is_floating_point()) for x in inputs]
    ref.train()
    out_r = ref(*ins_r)
    w = torch.randn(out_r.shape, generator=torch.Generator().manual_seed(seed))
    (out_r * w).sum().backward()
    ins_m = [x.to(cuda).requires_grad_(x.is_floating_point()) for x in inputs]
    mine.train()
    out_m = mine(*ins_m)
    (out_m * w.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert out_m.shape == out_r.shape
    assert G.max_rel(out_m.detach().float().cpu(), out_r.detach()) < tol_o
    G.check_grads(dict(mine.named_parameters()), {k: p.grad for k, p in ref.named_parameters()}, tol_g, cancelled)
    for xm, xr in zip(ins_m, ins_r):
        if input_grads and xr.grad is not None:  # (the encoders do not differentiate with respect to the clip)
            assert G.max_rel(xm.grad.cpu(), xr.grad) < tol_g
    return out_r.detach(), out_m.detach().float().cpu()


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_decoder_rnn_28_steps(cuda, dt):
    """DecoderRNN as predict.py:63-64 builds it (512 -> 3 x 512 -> 256 -> 101) over 28 steps, 4 clips."""
    from oracle import torch_oracle as O
    from video_classification_amd.modules import DecoderRNN
    torch.manual_seed(31)
    ref = O.DecoderRNN(512, 3, 512, 256, 0.0, 101)
    x = torch.randn(4, 28, 512, generator=torch.Generator().manual_seed(32))
    _fwd_bwd(ref, DecoderRNN(512, 3, 512, 256, 0.0, 101), [x], cuda, dt)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_crnn_28_frames(cuda, dt):
    """CRNNModel (EncoderCNN + DecoderRNN, tiny widths) on clips of 28 frames: 28 BatchNorm groups of B rows, whose
    running statistics receive 28 momentum updates in order (num_batches_tracked += 28)."""
    from oracle import torch_oracle as O
    from video_classification_amd.modules import CRNNModel
    torch.manual_seed(33)
    ref = O.CRNNModel(G.tiny_crnn_config())
    x = torch.randn(3, 28, 3, 64, 64, generator=torch.Generator().manual_seed(34))
    mine = CRNNModel(G.tiny_crnn_config())
    # convolution biases straight in front of a BatchNorm: their true gradient is zero (golden_util.check_grads)
    _fwd_bwd(ref, mine, [x], cuda, dt, cancelled=[f"encoder.conv{i}.0.bias" for i in (1, 2, 3, 4)],
             input_grads=False)
    tol_o = TOL[dt][0]
    bufs_r = dict(ref.named_buffers())
    for k, b in mine.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(bufs_r[k]) == 28, k
        else:
            assert G.max_rel(b.cpu(), bufs_r[k]) < tol_o, k


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_tcg_lstm_standalone_configuration_28_frames(cuda, dt):
    """TCG_LSTM(embed_dim=512, hidden=512, layers=3, k_tokens=8, down=4) as the stand-alone clusterSwin script builds
    it, on 28 frames: the long branch runs 7 steps and is interpolated 7 -> 28."""
    from oracle import torch_oracle as O
    from video_classification_amd.modules import TCG_LSTM
    torch.manual_seed(35)
    ref = O.TCG_LSTM(512, 512, 3, 8, 4, 101, 0.0)
    gen = torch.Generator().manual_seed(36)
    g = torch.randn(2, 28, 512, generator=gen)
    c = g.unsqueeze(2).repeat(1, 1, 8, 1) + 0.1 * torch.randn(2, 28, 8, 512, generator=gen)
    _fwd_bwd(ref, TCG_LSTM(512, 512, 3, 8, 4, 101, 0.0), [g, c], cuda, dt)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_predict_topk_on_a_28_frame_clip(cuda, dt):
    """predict.py:55-64, 134-142: ONE clip of 28 frames of 224 x 224 through encoder + decoder in eval mode, softmax,
    top-5 (full-width CRNN: 1024 / 768 / 512, 3 x 512 LSTM).  Indices equal the oracle's wherever its neighbouring
    probabilities differ by more than the output bound."""
    from oracle import torch_oracle as O
    from video_classification_amd import trainer as TR
    from video_classification_amd.modules import CRNNModel
    torch.manual_seed(37)
    ref = O.CRNNModel(G.full_crnn_config())
    with torch.no_grad():  # running statistics as after some training, so that eval-mode BatchNorm does something
        for k, b in ref.named_buffers():
            if k.endswith("running_mean"):
                b.normal_(0.0, 0.1)
            elif k.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    mine = CRNNModel(G.full_crnn_config())
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(cuda).set_compute_dtype(dt)
    x = torch.randn(1, 28, 3, 224, 224, generator=torch.Generator().manual_seed(38))
    ref.eval()
    with torch.no_grad():
        pr = torch.softmax(ref(x), dim=1)[0]
    p, idx = TR.predict_topk(mine, x.to(cuda), k=5)
    tol_o = TOL[dt][0]
    pr_sorted, idx_r = torch.sort(pr, descending=True, stable=True)
    assert G.max_rel(p, pr[idx]) < tol_o
    gaps = pr_sorted[:5] - pr_sorted[1:6]
    safe = gaps > 2 * tol_o * pr_sorted[0]
    print(f"\n  top-5 ranks compared: {int(safe.sum())} of 5")
    assert safe.sum().item() >= 1 or dt == torch.bfloat16
    assert torch.equal(idx[safe], idx_r[:5][safe])
