"""The data-parallel training step of the HIP path with two ranks on ONE GPU (collectives on gloo; see
tests/dp_worker.py): what the kernels' own gradient-ready notifications, the bucket all-reduces, the optimizer's wait
and its 1/world factor produce, in the eager executor and in the phased hipGraph executor, against a single process
that computes both replicas' gradients one after the other, sums them and steps with the same factor."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(mode, steps, prefix):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VC_SINGLE_DEVICE="1", VC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), mode, str(steps), prefix],
                                      env=env))
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [torch.load(f"{prefix}.rank{r}.pt", weights_only=False) for r in range(2)]


def _single_process_reference(cuda, steps, warm):
    """Both replicas' gradients from one model state, summed, one optimizer step with grad_scale = 1/2.  SGD with
    momentum: the update is linear in the gradient (under Adam the parameters whose true gradient is zero -- biases in
    front of a BatchNorm -- move by +-lr on rounding noise, differently in every run)."""
    sys.path.insert(0, HERE)
    import dp_worker as W
    from video_classification_amd import trainer as TR
    model = W.build(cuda)
    opt = TR.get_optimizer(model, "sgd", lr=1e-2, weight_decay=1e-4)
    opt.grad_scale = 0.5
    crit = TR.FusedCrossEntropyLoss()
    model.train()
    start = opt.flat.data.detach().cpu().clone()

    def buffers():
        return {k: v for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    for s in [0] * warm + list(range(steps)):
        total = torch.zeros_like(opt.flat.grad)
        before = {k: v.clone() for k, v in buffers().items()}
        for rank in (1, 0):
            if rank == 0:  # rank 0's running statistics see rank 0's clips only
                for k, v in buffers().items():
                    v.copy_(before[k])
            x, y = W.batch(rank, s, cuda)
            opt.zero_grad()
            crit(model(x), y).backward()
            total += opt.flat.grad
        opt.flat.grad.copy_(total)
        opt.step()
    torch.cuda.synchronize()
    return opt.flat.data.detach().cpu(), model.encoder.resnet[1].running_mean.detach().cpu(), start


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_two_ranks_match_single_process_sum(cuda, tmp_path, mode):
    steps = 3
    out = _launch(mode, steps, str(tmp_path / mode))
    # replicas stay identical
    assert torch.equal(out[0]["params"], out[1]["params"])
    # every bucket's all-reduce was started from the backward pass itself (eager) / after the phase that completed it
    assert out[0]["buckets"] >= 3
    assert out[0]["late"] == [0] * steps and out[1]["late"] == [0] * steps
    if mode == "graph":
        assert out[0]["phases"] == 4                                  # three stage boundaries in the trunk
        assert sum(len(b) for b in out[0]["phase_buckets"]) == out[0]["buckets"]
        assert len(out[0]["phase_buckets"][0]) >= 1                   # something is exchanged beside the later phases
    ref_params, ref_rm, start = _single_process_reference(cuda, steps, warm=1 if mode == "graph" else 0)
    moved = (ref_params - start).abs().max().item()
    assert moved > 1e-3, moved                      # the steps did change the weights
    d = (out[0]["params"] - ref_params).abs().max().item()
    assert d < 1e-2 * moved, (d, moved)             # fp32 path: summation order of the atomics (measured 3e-3)
    assert (out[0]["running_mean"] - ref_rm).abs().max().item() < 1e-5


def _launch_one(mode, steps, prefix, **env_extra):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("VC_DIST_BACKEND", "VC_DP_FORCE", "VC_SINGLE_DEVICE")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", VC_WORKER_DET="1", **env_extra)
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), mode, str(steps), prefix], env=env)
    try:
        assert p.wait(timeout=240) == 0
    finally:
        if p.poll() is None:
            p.kill()
    return torch.load(f"{prefix}.rank0.pt", weights_only=False)


@pytest.mark.timeout(400)
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_one_rank_on_rccl_matches_plain_step(cuda, tmp_path, mode):
    """The exchange path on its REAL backend: a one-rank RCCL ("nccl") communicator with the collectives forced on
    (VC_DP_FORCE=1) -- ncclBroadcast of the parameters, one async ncclAllReduce per gradient bucket on the side stream
    launched from the kernels' gradient-ready notifications (eager) or between the phase replays (graph), the stream
    fences, the optimizer's wait.  A sum over one rank is the identity, so with ordered sums the weights must equal
    those of the plain single-rank step bit for bit (the reference's counterpart is nn.DataParallel, train.py:515-518)."""
    steps = 3
    out = _launch_one(mode, steps, str(tmp_path / (mode + "_rccl")), VC_DP_FORCE="1", VC_DIST_BACKEND="nccl")
    ref = _launch_one(mode, steps, str(tmp_path / (mode + "_plain")), VC_WORKER_PLAIN="1")
    assert out["backend"] == "nccl" and ref["backend"] is None
    assert out["buckets"] >= 3
    warm = 1 if mode == "graph" else 0
    assert out["launches"] == out["buckets"] * (steps + warm), (out["launches"], out["buckets"])
    assert out["late"] == [0] * steps
    assert out["side_stream_work"]
    if mode == "graph":
        assert out["phases"] == 4 and sum(len(b) for b in out["phase_buckets"]) == out["buckets"]
    assert torch.equal(out["params"], ref["params"])
    assert torch.equal(out["running_mean"], ref["running_mean"])
