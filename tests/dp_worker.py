"""One rank of the data-parallel rehearsal that tests/test_gpu_dp.py starts (not a test module itself).

    RANK / WORLD_SIZE / MASTER_* set by the parent; VC_SINGLE_DEVICE=1 VC_DIST_BACKEND=gloo put every rank on GPU 0
    with the collectives on gloo (RCCL refuses two ranks on one device), so the whole multi-rank flow -- broadcast,
    gradient-ready notifications from the kernels' autograd functions, bucket launches, the wait in the optimizer,
    the 1/world factor -- runs on a one-GPU box.

    With WORLD_SIZE=1 VC_DP_FORCE=1 VC_DIST_BACKEND=nccl the same flow runs on a one-rank RCCL communicator (broadcast,
    side-stream all-reduces with async handles, stream fences, the phased-graph interleave on the real backend);
    VC_WORKER_PLAIN=1 runs the plain single-rank step instead (no process group, no bucketer) and VC_WORKER_DET=1
    switches the ordered sums on, so that the two can be compared bit for bit.

    python tests/dp_worker.py <mode: eager|graph> <steps> <out file prefix>
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch  # noqa: E402


def build(dev):
    from video_classification_amd import modules as M
    M.RESNET_LAYERS.setdefault("tiny", (1, 1, 1, 1))
    cfg = types.SimpleNamespace(CNN_fc_hidden1=64, CNN_fc_hidden2=48, CNN_embed_dim=32, dropout_p=0.0,
                                use_pretrained=False, RNN_hidden_layers=2, RNN_hidden_nodes=32, RNN_FC_dim=16,
                                num_classes=101, arch="tiny")
    torch.manual_seed(21)
    return M.ResNetCRNNModel(cfg).to(dev).set_compute_dtype(torch.float32)


def batch(rank, step, dev):
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    return torch.randn(4, 2, 3, 64, 64, generator=g).to(dev), torch.randint(0, 101, (4, 1), generator=g).to(dev)


def main():
    mode, steps, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    from video_classification_amd import ops
    from video_classification_amd import trainer as TR
    plain = os.environ.get("VC_WORKER_PLAIN", "0") == "1"
    rank, local_rank, world = (0, 0, 1) if plain else TR.init_distributed()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if os.environ.get("VC_WORKER_DET", "0") == "1":
        ops.set_deterministic(True, workspace_mb=64)
    model = build(dev)
    if rank != 0:  # rank 0's weights must win: start the others from something else
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    opt = TR.get_optimizer(model, "sgd", lr=1e-2, weight_decay=1e-4)  # linear in the gradient (see the test)
    # small buckets: several per phase of the backward pass
    import torch.distributed as dist
    if plain:
        bucketer = types.SimpleNamespace(buckets=[], late=0, launches=0)
    else:
        bucketer = TR.make_data_parallel(model, opt, bucket_mb=1)
    crit = TR.FusedCrossEntropyLoss()
    model.train()
    info = {"buckets": len(bucketer.buckets), "late": [],
            "backend": dist.get_backend() if dist.is_initialized() else None}
    if mode == "graph":
        from video_classification_amd.graph import GraphedTrainStep
        x0, y0 = batch(rank, 0, dev)
        # the warm-up steps of the capture are real training steps: replay them in the single-process check too
        step = GraphedTrainStep(model, opt, crit, x0, y0, warmup=1)
        info["phases"] = len(step.graphs)
        info["phase_buckets"] = step.phase_buckets
        for s in range(steps):
            x, y = batch(rank, s, dev)
            step(x, y)
            info["late"].append(bucketer.late)
    else:
        for s in range(steps):
            x, y = batch(rank, s, dev)
            opt.zero_grad()
            crit(model(x), y).backward()
            opt.step()
            info["late"].append(bucketer.late)
    torch.cuda.synchronize()
    ops.check_lstm_error()
    info["launches"] = bucketer.launches
    info["side_stream_work"] = bool(ops.SIDE_STREAM_WORK)
    info["params"] = opt.flat.data.detach().cpu()
    info["running_mean"] = model.encoder.resnet[1].running_mean.detach().cpu()
    torch.save(info, f"{prefix}.rank{rank}.pt")
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
