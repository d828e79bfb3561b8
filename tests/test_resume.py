"""Checkpoint / resume of the trainer (reference: utils/common/training_utils.py:199-243, train.py:686-700,
848-873).  The host logic is checked here on the CPU with torch's own optimizer; the GPU test continues a
training run of the HIP path from a checkpoint and holds it to the uninterrupted run."""
import os

import pytest
import torch
import torch.nn as nn

from video_classification_amd import trainer


def _tiny():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(6, 8), nn.ReLU(), nn.Linear(8, 3))


def _epoch(model, opt):
    x = torch.linspace(-1, 1, 24).reshape(4, 6)
    opt.zero_grad()
    model(x).square().mean().backward()
    opt.step()


def _run(tmp_path, sched_name, total, stop, with_sched_state, reference_layout):
    """lr after every epoch of an uninterrupted run, and of a run that stops after `stop` epochs and resumes."""
    def fresh():
        m = _tiny()
        o = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-4)
        s = trainer.get_scheduler(o, sched_name, step_size=3, gamma=0.5)
        return m, o, s

    def step_sched(s, loss):
        s.step(loss) if isinstance(s, torch.optim.lr_scheduler.ReduceLROnPlateau) else s.step()

    m, o, s = fresh()
    straight = []
    for e in range(total):
        _epoch(m, o)
        step_sched(s, 1.0)
        straight.append(o.param_groups[0]["lr"])
    w_straight = [p.detach().clone() for p in m.parameters()]

    m, o, s = fresh()
    resumed = []
    ckdir = str(tmp_path / "ckpt")
    for e in range(stop):
        _epoch(m, o)
        step_sched(s, 1.0)
        resumed.append(o.param_groups[0]["lr"])
    path = trainer.checkpoint_path_for_epoch(ckdir, stop)
    if reference_layout:  # exactly the dictionary the reference writes, under DataParallel
        os.makedirs(ckdir, exist_ok=True)
        torch.save({"epoch": stop, "model_state_dict": {"module." + k: v for k, v in m.state_dict().items()},
                    "optimizer_state_dict": o.state_dict(), "loss": 1.0}, path)
    else:
        trainer.save_checkpoint(m, o, stop, 1.0, path, scheduler=s if with_sched_state else None)
    m, o, s = fresh()
    for p in m.parameters():
        p.data.add_(1.0)  # must be overwritten by the restore
    start = trainer.resume_training(m, o, s, ckdir, stop)
    assert start == stop + 1
    for e in range(stop, total):
        _epoch(m, o)
        step_sched(s, 1.0)
        resumed.append(o.param_groups[0]["lr"])
    return straight, resumed, w_straight, [p.detach().clone() for p in m.parameters()]


@pytest.mark.parametrize("sched", ["step", "cosine"])
@pytest.mark.parametrize("layout", ["own+scheduler", "own", "reference"])
def test_resume_matches_uninterrupted_run(tmp_path, sched, layout):
    straight, resumed, w0, w1 = _run(tmp_path, sched, total=8, stop=4, with_sched_state=layout == "own+scheduler",
                                     reference_layout=layout == "reference")
    assert resumed == pytest.approx(straight, rel=1e-12, abs=0)
    assert len(set(straight)) > 2  # the schedule did move across the resume point
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)


def test_resume_plateau_keeps_learning_rate(tmp_path):
    straight, resumed, _, _ = _run(tmp_path, "plateau", total=10, stop=6, with_sched_state=True,
                                   reference_layout=False)
    assert resumed == pytest.approx(straight, rel=1e-12, abs=0)
    # without the scheduler's state the learning rate of the checkpoint is kept, the patience window restarts
    straight, resumed, _, _ = _run(tmp_path, "plateau", total=6, stop=5, with_sched_state=False,
                                   reference_layout=True)
    assert resumed[5] == resumed[4] == straight[4]


def test_resume_names_available_checkpoints(tmp_path):
    m = _tiny()
    o = torch.optim.Adam(m.parameters(), lr=1e-2)
    trainer.save_checkpoint(m, o, 3, 0.5, trainer.checkpoint_path_for_epoch(str(tmp_path), 3))
    with pytest.raises(FileNotFoundError, match="checkpoint_epoch_3.pth"):
        trainer.resume_training(m, o, None, str(tmp_path), 7)
    assert trainer.load_checkpoint(m, o, str(tmp_path / "nothing.pth")) == 1


@pytest.mark.gpu
def test_resume_hip_training_run(cuda, tmp_path):
    """Three optimizer steps, checkpoint, two more -- against a fresh process state that loads the checkpoint and
    takes the same two steps: same learning rates, weights within the run-to-run noise of the fp32 atomics."""
    from types import SimpleNamespace

    from video_classification_amd import modules as M

    def fresh():
        torch.manual_seed(3)
        model = M.Conv3DModel(SimpleNamespace(dropout_p=0.0, num_classes=11)).to(cuda)
        opt = trainer.get_optimizer(model, "adam", lr=1e-3, weight_decay=1e-4)
        sch = trainer.get_scheduler(opt, "step", step_size=2, gamma=0.1)
        return model, opt, sch

    g = torch.Generator().manual_seed(5)
    clips = [torch.randn(4, 1, 4, 16, 16, generator=g).to(cuda) for _ in range(5)]
    labels = [torch.randint(0, 11, (4, 1), generator=g).to(cuda) for _ in range(5)]
    crit = trainer.FusedCrossEntropyLoss()

    def one(model, opt, sch, i):
        model.train()
        opt.zero_grad()
        loss = crit(model(clips[i]), labels[i].squeeze())
        loss.backward()
        opt.step()
        sch.step()
        return float(loss.detach()), opt.param_groups[0]["lr"]

    model, opt, sch = fresh()
    for i in range(3):
        one(model, opt, sch, i)
    trainer.save_checkpoint(model, opt, 3, 0.0, trainer.checkpoint_path_for_epoch(str(tmp_path), 3), scheduler=sch)
    tail_a = [one(model, opt, sch, i) for i in (3, 4)]
    w_a = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}

    model, opt, sch = fresh()
    assert trainer.resume_training(model, opt, sch, str(tmp_path), 3) == 4
    assert opt.steps == 3 and int(opt.step_dev.item()) == 3
    assert float(opt.lr_dev.item()) == pytest.approx(opt.param_groups[0]["lr"])
    tail_b = [one(model, opt, sch, i) for i in (3, 4)]
    w_b = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}

    for (la, lra), (lb, lrb) in zip(tail_a, tail_b):
        assert lra == pytest.approx(lrb, rel=1e-12)
        assert la == pytest.approx(lb, rel=2e-3)
    for k in w_a:  # a step of Adam moves an entry by at most lr = 1e-3; the two runs differ by a small part of that
        d = (w_a[k] - w_b[k]).abs().max()
        assert d <= 1e-4 + 1e-3 * w_a[k].abs().max(), (k, float(d))
