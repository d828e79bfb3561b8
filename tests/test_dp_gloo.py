"""Data-parallel plumbing on CPU with the gloo backend, world sizes 2 and 4 (the N>1 path of bench.py uses the same
classes over RCCL): flat parameter/gradient buffers, bucket partition, gradient-ready notifications, the
overlapped all-reduce and its finish(), initial broadcast.

Correctness criterion: gradients summed over ranks and scaled by 1/world equal the
single-process gradients of the concatenated batch (a BatchNorm-free model, mean cross-entropy)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import torch_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return O.DecoderRNN(CNN_embed_dim=16, h_RNN_layers=2, h_RNN=24, h_FC_dim=8, drop_p=0.0, num_classes=11)


def _worker(rank, world, port, bucket_bytes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from video_classification_amd import trainer as TR
    r, lr, w = TR.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    model = _model()
    if rank != 0:  # replicas start different: the broadcast must make them equal to rank 0
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    flat = TR.FlatParams(model)
    # parameters now live in the flat buffer and their grads are views of the flat gradient buffer
    for i, p in enumerate(flat.params):
        assert p.data_ptr() == flat.data[flat.offsets[i]:].data_ptr()
        assert p.grad.data_ptr() == flat.grad[flat.offsets[i]:].data_ptr()

    class _Opt:  # the two attributes make_data_parallel needs from the fused optimizer
        pass
    opt = _Opt()
    opt.flat = flat
    bucketer = TR.make_data_parallel(model, opt, bucket_mb=1)
    bucketer = TR.GradBucketer(flat, bucket_bytes=bucket_bytes)
    assert opt.grad_scale == 1.0 / world
    ref = _model()
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a, b)  # broadcast from rank 0

    g = torch.Generator().manual_seed(42)
    x = torch.randn(3 * world, 5, 16, generator=g)
    y = torch.randint(0, 11, (3 * world,), generator=g)
    xs, ys = x[rank * 3:(rank + 1) * 3], y[rank * 3:(rank + 1) * 3]
    flat.zero_grad()
    bucketer.reset()
    loss = nn.CrossEntropyLoss()(model(xs), ys)
    grads = torch.autograd.grad(loss, flat.params)
    # backward produces gradients in reverse registration order; signal each as it is written
    for p, gr in reversed(list(zip(flat.params, grads))):
        p.grad.copy_(gr)
        bucketer.notify(p)
    launched_early = sum(bucketer.launched)
    bucketer.finish()
    flat.grad.mul_(opt.grad_scale)

    loss_full = nn.CrossEntropyLoss()(ref(x), y)
    gfull = torch.autograd.grad(loss_full, list(ref.parameters()))
    err = max((p.grad - gf).abs().max().item() for p, gf in zip(flat.params, gfull))
    q.put((rank, err, len(bucketer.buckets), launched_early))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bucket_bytes", [(2, 1 << 30), (2, 4096), (4, 4096)])
def test_bucketed_allreduce_matches_full_batch(world, bucket_bytes):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, nb, early in res:
        assert err < 1e-6, (rank, err)
        if bucket_bytes == 4096:
            assert nb > 1           # several buckets ...
            assert early == nb      # ... each launched by its last notify, before finish()
        else:
            assert nb == 1


def test_bucket_partition_covers_flat_buffer():
    from video_classification_amd import trainer as TR
    model = _model()
    flat = TR.FlatParams(model)
    b = TR.GradBucketer(flat, bucket_bytes=2048)
    spans = sorted(b.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == flat.total
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0
    assert sum(b.sizes) == len(flat.params)
    # state_dict keys are unchanged by the re-homing (no wrapper, no `module.` prefix)
    assert list(model.state_dict().keys()) == list(_model().state_dict().keys())


def test_strip_module_prefix():
    from video_classification_amd import trainer as TR
    sd = {"module.encoder.fc1.weight": 1, "decoder.fc2.bias": 2}
    assert TR.strip_module_prefix(sd) == {"encoder.fc1.weight": 1, "decoder.fc2.bias": 2}
