"""GPU parity of the bandwidth-bound kernels (BatchNorm with groups, pooling, LayerNorm, cross-entropy,
fused optimizers, dropout) against plain PyTorch fp32 on the CPU, through the autograd wrappers / C ABI.

Tolerances: float32 path 2e-5 (relative to the largest magnitude), bfloat16 path 2e-2 (storage rounding).
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import golden_util as G

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]


def tol(dt):
    return 2e-5 if dt == torch.float32 else 2e-2


def rnd(dt, *shape, seed=0):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
    return x.bfloat16().float() if dt == torch.bfloat16 else x


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("shape,groups", [((6, 1, 5, 7, 24), 3), ((8, 1, 1, 1, 96), 2), ((4, 3, 6, 6, 32), 1),
                                          ((6, 1, 3, 3, 2304), 2)])
@pytest.mark.parametrize("relu,res,inplace", [(True, False, False), (True, True, False), (True, True, True),
                                              (False, False, False)])
def test_batchnorm_groups(cuda, dt, shape, groups, relu, res, inplace):
    """Training-mode BN over G groups of rows == the reference's per-time-step loop: G separate BN calls in
    order, running stats updated G times.  `inplace`: the backward pass may overwrite the incoming gradient
    with its ReLU-masked form (the residual's gradient), as the bottleneck blocks ask for."""
    from video_classification_amd import ops
    C = shape[-1]
    x = rnd(dt, *shape, seed=1) * 2 + 0.5
    if dt == torch.bfloat16:
        # keep the input exactly representable: a rounded input would flip ReLU decisions of outputs within
        # 2^-9 of zero, and each flip changes that element's gradient by O(1)
        x = x.bfloat16().float()
    r = rnd(dt, *shape, seed=2) if res else None
    dy = rnd(dt, *shape, seed=3)
    bn_ref = nn.BatchNorm1d(C, momentum=0.01)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.randn(C, generator=torch.Generator().manual_seed(4)))
        bn_ref.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(5)))
    bn_mine = nn.BatchNorm1d(C, momentum=0.01)
    bn_mine.load_state_dict(bn_ref.state_dict())
    bn_mine.to(cuda)
    # reference: per-group BN on [rows, C]
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    n = shape[0] // groups
    outs = []
    for g in range(groups):
        o = bn_ref(xr[g * n:(g + 1) * n].reshape(-1, C))
        if res:
            o = o + rr[g * n:(g + 1) * n].reshape(-1, C)
        outs.append(F.relu(o) if relu else o)
    out_r = torch.cat(outs).reshape(shape)
    out_r.backward(dy)
    xm = x.to(cuda, dt).requires_grad_(True)
    rm = r.to(cuda, dt).requires_grad_(True) if res else None
    out_m = ops.batch_norm(xm, bn_mine, groups, relu=relu, residual=rm, grad_inplace=inplace)
    out_m.backward(dy.to(cuda, dt))
    t = tol(dt)
    assert G.max_rel(out_m.float().cpu(), out_r.detach()) < t
    assert G.max_rel(xm.grad.float().cpu(), xr.grad) < 2 * t
    if res:
        assert G.max_rel(rm.grad.float().cpu(), rr.grad) < t
    assert G.max_rel(bn_mine.weight.grad.cpu(), bn_ref.weight.grad) < 2 * t
    assert G.max_rel(bn_mine.bias.grad.cpu(), bn_ref.bias.grad) < 2 * t
    assert G.max_rel(bn_mine.running_mean.cpu(), bn_ref.running_mean) < t
    assert G.max_rel(bn_mine.running_var.cpu(), bn_ref.running_var) < t
    assert int(bn_mine.num_batches_tracked) == groups
    # eval mode uses the running statistics
    bn_ref.eval()
    bn_mine.eval()
    with torch.no_grad():
        oe_r = bn_ref(x.reshape(-1, C)).reshape(shape)
        oe_m = ops.batch_norm(x.to(cuda, dt), bn_mine, groups)
    assert G.max_rel(oe_m.float().cpu(), oe_r) < t


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("k,s,p,N,D,H,W,C,groups", [(3, 2, 1, 6, 1, 14, 14, 64, 3), (3, 2, 1, 4, 1, 9, 11, 24, 2),
                                                    (2, 2, 0, 2, 4, 8, 6, 32, 1), (2, 2, 0, 2, 3, 7, 7, 16, 1)])
def test_batchnorm_relu_maxpool_fused(cuda, dt, k, s, p, N, D, H, W, C, groups):
    """BatchNorm + ReLU + max-pool in one pass each way (vc_bn_pool_fwd / vc_bn_pool_bwd) against the two-pass form
    of the same library (BatchNorm + ReLU, then max-pool): identical pooled output and running statistics, input
    and parameter gradients equal up to the order of the reductions; and against plain PyTorch in fp32."""
    from video_classification_amd import ops
    x = rnd(dt, N, D, H, W, C, seed=31) * 1.5 + 0.2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = rnd(dt, N, D, Ho, Wo, C, seed=32)
    res = {}
    for mode in (2, 0):  # every geometry fused / never fused
        ops.FUSE_BN_POOL = mode
        try:
            bn = nn.BatchNorm1d(C, momentum=0.01)
            with torch.no_grad():
                bn.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(33)) + 0.5)
                bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(34)) * 0.3)
            bn.to(cuda)
            xm = x.to(cuda, dt).requires_grad_(True)
            out = ops.batch_norm_relu_pool(xm, bn, groups, k, s, p)
            out.backward(dy.to(cuda, dt))
            torch.cuda.synchronize()
            res[mode] = (out.detach().float().cpu(), xm.grad.float().cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu(),
                         bn.running_mean.cpu(), bn.running_var.cpu(), int(bn.num_batches_tracked))
        finally:
            ops.FUSE_BN_POOL = 1
    f, u = res[2], res[0]
    assert torch.equal(f[0], u[0])
    t = tol(dt)
    for i in (1, 2, 3):
        assert G.max_rel(f[i], u[i]) < t, i
    assert torch.equal(f[4], u[4]) and torch.equal(f[5], u[5]) and f[6] == u[6] == groups
    if dt == torch.float32:  # plain PyTorch: per-group BatchNorm, ReLU, per-frame max-pool
        xr = x.clone().requires_grad_(True)
        bnr = nn.BatchNorm1d(C, momentum=0.01)
        with torch.no_grad():
            bnr.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(33)) + 0.5)
            bnr.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(34)) * 0.3)
        fr = xr.view(N * D, H, W, C)
        npg = N * D // groups
        a = torch.cat([F.relu(bnr(fr[g * npg:(g + 1) * npg].reshape(-1, C))).view(npg, H, W, C) for g in range(groups)])
        o = F.max_pool2d(a.permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1).reshape(N, D, Ho, Wo, C)
        o.backward(dy)
        assert G.max_rel(f[0], o.detach()) < t
        assert G.max_rel(f[1], xr.grad) < 2 * t
        assert G.max_rel(f[2], bnr.weight.grad) < 2 * t and G.max_rel(f[3], bnr.bias.grad) < 2 * t


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("k,s,p,H,W", [(3, 2, 1, 14, 14), (3, 2, 1, 24, 2), (3, 2, 1, 2, 20), (3, 2, 1, 9, 11), (2, 2, 0, 8, 6),
                                       (2, 2, 0, 7, 7)])
def test_maxpool(cuda, dt, k, s, p, H, W):
    from video_classification_amd import ops
    x = F.relu(rnd(dt, 3, 16, H, W, seed=7))  # post-ReLU input: many exact ties at zero
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, k, s, p)
    dy = rnd(dt, *yr.shape, seed=8)
    yr.backward(dy)
    xm = x.permute(0, 2, 3, 1).contiguous().to(cuda, dt).requires_grad_(True)
    ym = ops.max_pool2d(xm, k, s, p)
    ym.backward(dy.permute(0, 2, 3, 1).contiguous().to(cuda, dt))
    assert torch.equal(ym.float().cpu().permute(0, 3, 1, 2), yr.detach())
    # ties at zero may pick different window positions; the ReLU in front zeroes those gradients anyway
    mask = (x > 0).float()
    assert G.max_rel(xm.grad.float().cpu().permute(0, 3, 1, 2) * mask, xr.grad * mask) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("S", [49, 784, 300], ids=["resnet_7x7", "conv3d_16x7x7", "ragged_300"])
def test_global_avgpool(cuda, dt, S):
    from video_classification_amd import ops
    x = rnd(dt, 5, S, 64, seed=9)
    xr = x.clone().requires_grad_(True)
    yr = xr.mean(1)
    dy = rnd(dt, 5, 64, seed=10)
    yr.backward(dy)
    xm = x.to(cuda, dt).requires_grad_(True)
    ym = ops.global_avg_pool(xm)
    ym.backward(dy.to(cuda, dt))
    assert G.max_rel(ym.float().cpu(), yr.detach()) < tol(dt)
    assert G.max_rel(xm.grad.float().cpu(), xr.grad) < tol(dt)


@pytest.mark.parametrize("B,C", [(4, 101), (37, 101), (3, 7)])
def test_cross_entropy_head(cuda, B, C):
    from video_classification_amd import ops
    logits = torch.randn(B, 104, generator=torch.Generator().manual_seed(11)) * 3
    target = torch.randint(0, C, (B, 1), generator=torch.Generator().manual_seed(12))
    lr = logits[:, :C].clone().requires_grad_(True)
    loss_r = F.cross_entropy(lr, target.squeeze(1))
    loss_r.backward()
    lm = logits.to(cuda).requires_grad_(True)
    stats = {}
    loss_m = ops.cross_entropy(lm[:, :C], target.to(cuda), stats)
    loss_m.backward()
    assert abs(loss_m.item() - loss_r.item()) < 1e-5
    assert G.max_rel(lm.grad[:, :C].cpu(), lr.grad) < 1e-5
    assert torch.all(lm.grad[:, C:] == 0)
    assert torch.equal(stats["pred"].cpu(), lr.argmax(1))
    assert int(stats["correct"]) == int((lr.argmax(1) == target.squeeze(1)).sum())


def test_argmax_ties_take_first_index(cuda):
    from video_classification_amd import ops
    logits = torch.zeros(2, 104)
    logits[0, 5] = logits[0, 70] = 2.0
    logits[1, 100] = logits[1, 3] = 1.0
    stats = {}
    ops.cross_entropy(logits.to(cuda)[:, :101], torch.tensor([5, 3]).to(cuda), stats)
    assert stats["pred"].tolist() == [5, 3]
    assert torch.equal(stats["pred"].cpu(), logits[:, :101].argmax(1))


@pytest.mark.parametrize("name,kw", [("adam", dict(lr=1e-3, weight_decay=1e-4)), ("adamw", dict(lr=1e-3, weight_decay=1e-2)),
                                     ("sgd", dict(lr=1e-2, weight_decay=1e-4))])
def test_fused_optimizers_match_torch(cuda, name, kw):
    """Three steps of the fused flat-buffer optimizer against torch.optim on the CPU (training_utils.py:324-329)."""
    from video_classification_amd import trainer as TR
    torch.manual_seed(0)
    ref = nn.Sequential(nn.Linear(13, 7), nn.Linear(7, 5))
    mine = nn.Sequential(nn.Linear(13, 7), nn.Linear(7, 5))
    mine.load_state_dict(ref.state_dict())
    mine.to(cuda)
    if name == "adam":
        opt_r = torch.optim.Adam(ref.parameters(), **kw)
    elif name == "adamw":
        opt_r = torch.optim.AdamW(ref.parameters(), **kw)
    else:
        opt_r = torch.optim.SGD(ref.parameters(), momentum=0.9, **kw)
    opt_m = TR.FusedOptimizer(mine, name, **kw)
    for step in range(3):
        opt_m.zero_grad()
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            g = torch.randn(pr.shape, generator=torch.Generator().manual_seed(100 + step))
            pr.grad = g.clone()
            pm.grad.copy_(g)
        opt_r.step()
        opt_m.step()
    for pr, pm in zip(ref.parameters(), mine.parameters()):
        assert (pm.detach().cpu() - pr.detach()).abs().max().item() < 2e-6
    # optimizer state has torch's layout (checkpoint compatibility)
    sd = opt_m.state_dict()
    assert set(sd.keys()) == {"state", "param_groups"}
    if name != "sgd":
        assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("name", ["adam", "sgd"])
def test_fused_optimizer_resumes_torch_state(cuda, name):
    """A torch.optim state_dict taken after two steps (momentum / moments in place) is loaded into a fresh fused
    optimizer; the third step must equal torch's third step.  One parameter never receives a gradient: torch keeps no
    state for it, and a caller's model.zero_grad() (set_to_none) detaches every .grad from the flat buffer."""
    from video_classification_amd import trainer as TR
    torch.manual_seed(0)
    ref = nn.Sequential(nn.Linear(13, 7), nn.Linear(7, 5), nn.Linear(5, 3))
    mine = nn.Sequential(nn.Linear(13, 7), nn.Linear(7, 5), nn.Linear(5, 3))
    kw = dict(lr=1e-2, weight_decay=1e-4)
    opt_r = torch.optim.SGD(ref.parameters(), momentum=0.9, **kw) if name == "sgd" else torch.optim.Adam(ref.parameters(), **kw)
    unused = {id(p) for p in ref[2].parameters()}

    def grads(model, step, to=None):
        for p, q in zip(ref.parameters(), model.parameters()):
            if id(p) in unused:
                continue
            g = torch.randn(p.shape, generator=torch.Generator().manual_seed(200 + step))
            q.grad = g.clone().to(to) if to is not None else g.clone()

    for step in range(2):
        opt_r.zero_grad()
        grads(ref, step)
        opt_r.step()
    mine.load_state_dict(ref.state_dict())
    mine.to(cuda)
    opt_m = TR.FusedOptimizer(mine, name, **kw)
    opt_m.load_state_dict(opt_r.state_dict())
    opt_r.zero_grad()
    grads(ref, 2)
    opt_r.step()
    mine.zero_grad()            # set_to_none: the parameters lose their views of the flat gradient buffer
    opt_m.flat.zero_grad()
    grads(mine, 2, to=cuda)     # ... and receive freshly allocated gradients
    opt_m.step()
    for (k, pr), pm in zip(ref.named_parameters(), mine.parameters()):
        if id(pr) in unused and name == "adam":
            continue  # never-touched parameters: torch skips them, the flat kernel applies its weight-decay term
        assert (pm.detach().cpu() - pr.detach()).abs().max().item() < 2e-6, k
    for i, p in enumerate(opt_m.flat.params):
        assert p.grad.data_ptr() == opt_m.flat.view_of(opt_m.flat.grad, i).data_ptr()


def test_fused_optimizer_parameter_without_gradient_after_set_to_none(cuda):
    """A parameter that received a gradient in step 1 and none in step 2, with the caller clearing gradients by
    model.zero_grad(set_to_none=True) (which does not touch the flat buffer): step 2 must see a zero gradient for it,
    not step 1's.  SGD without momentum or weight decay: such a parameter must not move in step 2."""
    from video_classification_amd import trainer as TR
    torch.manual_seed(3)
    net = nn.Sequential(nn.Linear(6, 4), nn.Linear(4, 2)).to(cuda)
    opt = TR.FusedOptimizer(net, "sgd", lr=0.1, momentum=0.0, weight_decay=0.0)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    before = [p.detach().clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    for p in net[0].parameters():  # only the first layer gets a gradient this time
        p.grad = torch.ones_like(p)
    opt.step()
    after = [p.detach() for p in net.parameters()]
    for i, (b, a) in enumerate(zip(before, after)):
        moved = (a - b).abs().max().item()
        assert (abs(moved - 0.1) < 1e-6) if i < 2 else (moved == 0.0), (i, moved)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_dropout_statistics_and_backward(cuda, dt):
    from video_classification_amd import ops
    x = torch.ones(64, 1024, device=cuda, dtype=dt, requires_grad=True)
    ops.manual_seed(1)
    y = ops.dropout(x, 0.3, True)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    assert torch.allclose(y[y != 0].float(), torch.tensor(1 / 0.7), rtol=1e-2)
    y.sum().backward()
    assert torch.equal(x.grad != 0, y != 0)
    y2 = ops.dropout(x, 0.3, True)
    assert not torch.equal(y2 != 0, y != 0)  # a new mask every call
    assert ops.dropout(x, 0.3, False) is x


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("M,C", [(50, 96), (7, 768), (300, 384)])
def test_layernorm(cuda, dt, M, C):
    from video_classification_amd import ops
    x = rnd(dt, M, C, seed=20) * 3 + 1
    dy = rnd(dt, M, C, seed=21)
    ln_r = nn.LayerNorm(C)
    with torch.no_grad():
        ln_r.weight.copy_(torch.randn(C, generator=torch.Generator().manual_seed(22)))
        ln_r.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(23)))
    ln_m = nn.LayerNorm(C)
    ln_m.load_state_dict(ln_r.state_dict())
    ln_m.to(cuda)
    xr = x.clone().requires_grad_(True)
    yr = ln_r(xr)
    yr.backward(dy)
    xm = x.to(cuda, dt).requires_grad_(True)
    ym = ops.layer_norm(xm, ln_m)
    ym.backward(dy.to(cuda, dt))
    t = tol(dt)
    assert G.max_rel(ym.float().cpu(), yr.detach()) < t
    assert G.max_rel(xm.grad.float().cpu(), xr.grad) < 2 * t
    assert G.max_rel(ln_m.weight.grad.cpu(), ln_r.weight.grad) < 2 * t
    assert G.max_rel(ln_m.bias.grad.cpu(), ln_r.bias.grad) < 2 * t
    # the fork form: x continues in the residual stream, its gradient is added inside the backward kernel
    ln_r.zero_grad()
    ln_m.zero_grad()
    dres = rnd(dt, M, C, seed=24)
    xr = x.clone().requires_grad_(True)
    (ln_r(xr) * dy + xr * dres).sum().backward()
    xm = x.to(cuda, dt).requires_grad_(True)
    xs, ym = ops.layer_norm_fork(xm, ln_m)
    assert torch.equal(xs, xm)
    torch.autograd.backward([ym, xs], [dy.to(cuda, dt), dres.to(cuda, dt)])
    assert G.max_rel(xm.grad.float().cpu(), xr.grad) < 2 * t
    assert G.max_rel(ln_m.weight.grad.cpu(), ln_r.weight.grad) < 2 * t
    # only one of the two outputs used downstream
    xm2 = x.to(cuda, dt).requires_grad_(True)
    xs2, _ = ops.layer_norm_fork(xm2, ln_m)
    xs2.backward(dres.to(cuda, dt))
    assert G.max_rel(xm2.grad.float().cpu(), dres) < 1e-6


@pytest.mark.parametrize("T,B,In,H,layers,bidir", [(16, 32, 512, 512, 2, False), (16, 8, 192, 512, 2, True),
                                                   (4, 8, 96, 256, 2, True), (5, 3, 24, 32, 1, False),
                                                   (7, 40, 64, 64, 1, True)])
def test_persistent_lstm_matches_stepwise(cuda, T, B, In, H, layers, bidir):
    """The persistent LSTM kernels (W_hh / W_hh^T in registers, h and the gate gradients exchanged in tagged
    granules) against the per-step GEMM + cell kernels on the same bf16 inputs, forward alone and forward +
    backward: same math, different accumulation order -> 2e-2 of the output scale (bf16 h feeds back T times),
    5e-2 of the gradient scale (input gradient, W_hh and W_ih gradients of the first layer)."""
    from video_classification_amd import ops
    g = torch.Generator().manual_seed(T * 1000 + H)
    dirs = 2 if bidir else 1
    params = []
    for l in range(layers):
        kin = In if l == 0 else H * dirs
        for _ in range(dirs):
            k = 1.0 / H ** 0.5
            params += [((torch.rand(4 * H, kin, generator=g) * 2 - 1) * k).to(cuda).requires_grad_(True),
                       ((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(cuda).requires_grad_(True),
                       ((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(cuda).requires_grad_(True),
                       ((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(cuda).requires_grad_(True)]
    x = torch.randn(T, B, In, generator=g).to(cuda, torch.bfloat16)
    from video_classification_amd import _lib as L
    assert L.query("vc_lstm_persist_supported", 1, B, H, dirs) == 1
    assert L.query("vc_lstm_persist_bwd_supported", 1, B, H, dirs) == 1
    outs = []
    try:
        for fwd, bwd in ((False, False), (True, False), (True, True)):
            ops.PERSISTENT_LSTM, ops.PERSISTENT_LSTM_BWD = fwd, bwd
            for p in params:
                p.grad = None
            xx = x.clone().requires_grad_(True)
            y = ops._LSTM.apply(xx, H, layers, bidir, 0.0, *params)
            # a loss that weights time steps and features unevenly (the decoders read the last step only)
            wgt = torch.linspace(0.2, 1.0, T, device=cuda).view(T, 1, 1)
            (y.float().pow(2) * wgt).sum().backward()
            torch.cuda.synchronize()
            outs.append((y.detach().float().cpu(), xx.grad.float().cpu(), params[1].grad.clone().cpu(),
                         params[0].grad.clone().cpu()))
    finally:
        ops.PERSISTENT_LSTM = ops.PERSISTENT_LSTM_BWD = True
    ops.check_lstm_error()
    scale = outs[0][0].abs().max().item()
    for o in outs[1:]:
        assert (outs[0][0] - o[0]).abs().max().item() < 2e-2 * scale
        for i in (1, 2, 3):
            assert (outs[0][i] - o[i]).abs().max().item() < 5e-2 * outs[0][i].abs().max().item(), i
    # same forward, backward recurrence per step vs persistent: only the summation order of dh differs
    for i in (1, 2, 3):
        assert (outs[1][i] - outs[2][i]).abs().max().item() < 2e-2 * outs[1][i].abs().max().item(), i


@pytest.mark.parametrize("M,C", [(5000, 96), (3000, 768)])
def test_layernorm_gradients_in_deterministic_mode(cuda, M, C):
    """ops.set_deterministic: the LayerNorm parameter gradients (sums over M rows, blocks otherwise combined with
    atomics) repeat bit for bit and agree with the default mode up to summation order."""
    from video_classification_amd import ops
    ln = nn.LayerNorm(C).to(cuda)
    x = torch.randn(M, C, generator=torch.Generator().manual_seed(1)).to(cuda, torch.bfloat16)
    dy = torch.randn(M, C, generator=torch.Generator().manual_seed(2)).to(cuda, torch.bfloat16)

    def grads():
        ln.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        ops.layer_norm(xi, ln).backward(dy)
        torch.cuda.synchronize()
        return ln.weight.grad.clone(), ln.bias.grad.clone(), xi.grad.clone()
    base = grads()
    ops.set_deterministic(True, workspace_mb=64)
    try:
        a, b = grads(), grads()
    finally:
        ops.set_deterministic(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    for u, v in zip(a, base):
        assert G.max_rel(u.float().cpu(), v.float().cpu()) < 1e-4


def test_batchnorm_single_value_per_channel_raises(cuda):
    """Training-mode BatchNorm over one value per channel (a batch of one clip through the BatchNorm1d heads) raises as
    torch's does, instead of writing an infinite running variance; eval mode takes it."""
    from video_classification_amd import ops
    bn = nn.BatchNorm1d(16).to(cuda)
    x = torch.randn(3, 16, device=cuda).bfloat16()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        ops.batch_norm(x, bn, 3)          # three groups (time steps) of one frame each
    ref = nn.BatchNorm1d(16)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        ref(torch.randn(1, 16))
    assert torch.isfinite(bn.running_var).all()
    bn.eval()
    with torch.no_grad():
        assert torch.isfinite(ops.batch_norm(x, bn, 3).float()).all()
