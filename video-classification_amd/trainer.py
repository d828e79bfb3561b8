"""Training-step plumbing around the HIP modules: flat parameter / gradient buffers, the fused
optimizer, bucketed data-parallel gradient all-reduce over RCCL, and drop-in ``train_epoch`` /
``validate_epoch`` / ``get_optimizer`` / ``get_scheduler`` / checkpoint helpers with the reference's
signatures (utils/common/training_utils.py:49-163, 199-243, 311-353).

One process drives one GPU.  Replicas hold full weights; each rank works on its own clips; the only
exchange step is the gradient all-reduce (the reference's ``nn.DataParallel``, train.py:515-518,
is replaced by this).  BatchNorm statistics stay per replica and per time step, as under
DataParallel.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import ops


# --------------------------------------------------------------------------------------
# flat buffers
# --------------------------------------------------------------------------------------
class FlatParams:
    """Re-homes every parameter of `model` into one fp32 buffer (and its gradient into a second one) so
    that the optimizer is one kernel launch and gradient buckets are plain slices.  Parameters keep
    their identity, shape and state_dict keys; only their storage moves."""

    ALIGN = 64  # elements: every parameter starts on a 256-byte boundary

    def __init__(self, model):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0].device
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            if p.dtype != torch.float32:
                raise TypeError("master parameters must be float32")
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
        ops.bump_weights_epoch()

    def zero_grad(self):
        self.grad.zero_()

    def view_of(self, buf, i):
        o, p = self.offsets[i], self.params[i]
        return buf[o:o + p.numel()].view(p.shape)


# --------------------------------------------------------------------------------------
# fused optimizer (torch.optim.Optimizer subclass so that torch LR schedulers and the reference's
# checkpoint format keep working)
# --------------------------------------------------------------------------------------
class FusedOptimizer(torch.optim.Optimizer):
    MODES = {"adam": 0, "adamw": 1, "sgd": 2}

    def __init__(self, model_or_flat, name="adam", lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 momentum=0.9):
        self.flat = model_or_flat if isinstance(model_or_flat, FlatParams) else FlatParams(model_or_flat)
        name = name.lower()
        if name not in self.MODES:
            raise ValueError(f"Unsupported optimizer: {name}")
        self.mode = self.MODES[name]
        if name == "adamw" and weight_decay is None:
            weight_decay = 1e-2
        defaults = dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps, momentum=momentum)
        super().__init__(self.flat.params, defaults)
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data) if self.mode != 2 else None
        self.steps = 0
        self.grad_scale = 1.0
        self.bucketer = None
        # step count and learning rate also live on the device: the kernel reads them at execution time, so a
        # captured graph of the step keeps working as they change
        dev = self.flat.data.device
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        self.lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
        self._lr_host = float(lr)
        for i, p in enumerate(self.flat.params):
            st = self.state[p]
            if self.mode == 2:
                st["momentum_buffer"] = self.flat.view_of(self.m, i)
            else:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = self.flat.view_of(self.m, i)
                st["exp_avg_sq"] = self.flat.view_of(self.v, i)

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()
        if self.bucketer is not None:
            self.bucketer.reset()

    def _attach_grads(self):
        """Every parameter's .grad must be its slice of the flat gradient buffer (the kernels accumulate there and
        the fused step reads the flat buffer).  A caller's ``model.zero_grad()`` with set_to_none detaches them; a
        gradient that then got allocated elsewhere is moved back, a missing one reads as zero."""
        flat = self.flat
        for i, p in enumerate(flat.params):
            view = flat.view_of(flat.grad, i)
            g = p.grad
            if g is None:
                view.zero_()  # no gradient this step: the slice still holds the previous step's
                p.grad = view
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
                p.grad = view

    @torch.no_grad()
    def step(self, closure=None):
        self._attach_grads()
        if self.bucketer is not None:
            self.bucketer.finish()
        self.sync_lr()
        self.launch()
        self.after_step()

    def launch(self):
        """Enqueue the kernels of one optimizer step (no host-side bookkeeping; graph-capturable)."""
        g = self.param_groups[0]
        # a state_dict written by torch.optim carries only that optimizer's own hyper-parameters
        b1, b2 = g.get("betas", (0.9, 0.999))
        if self.mode == 2:
            b1 = g.get("momentum", 0.9)
        L.call("vc_counter_add", L.ptr(self.step_dev), L.i64(1), L.stream())
        L.call("vc_optim_step", self.mode, L.ptr(self.flat.data), L.ptr(self.flat.grad), L.ptr(self.m), L.ptr(self.v),
               L.i64(self.flat.total), float(g["lr"]), float(b1), float(b2), float(g.get("eps", 1e-8)),
               float(g["weight_decay"]), 0, float(self.grad_scale), L.ptr(self.lr_dev), L.ptr(self.step_dev),
               L.stream())
        ops.advance_seed_counter(self.flat.data.device)

    def sync_lr(self):
        """Push the host-side learning rate (param_groups[0]['lr'], e.g. after a scheduler step) to the device."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self.lr_dev.fill_(lr)
            self._lr_host = lr

    def after_step(self):
        """Host-side bookkeeping of one optimizer step (also called after a graph replay of the step)."""
        self.steps += 1
        if self.mode != 2:
            for p in self.flat.params:
                self.state[p]["step"] += 1
        ops.bump_weights_epoch()

    def load_state_dict(self, state_dict):
        # torch re-creates the state tensors; copy them back into the flat buffers
        # A torch.optim checkpoint carries no state for a parameter that never received a gradient (and SGD state
        # has no step count at all): such moments are zero, which is what the flat buffers are set to.
        super().load_state_dict(state_dict)
        steps = 0
        for i, p in enumerate(self.flat.params):
            st = self.state[p]
            mv = self.flat.view_of(self.m, i)
            if self.mode == 2:
                buf = st.get("momentum_buffer")
                mv.copy_(buf) if buf is not None else mv.zero_()
                st["momentum_buffer"] = mv
            else:
                vv = self.flat.view_of(self.v, i)
                mv.copy_(st["exp_avg"]) if "exp_avg" in st else mv.zero_()
                vv.copy_(st["exp_avg_sq"]) if "exp_avg_sq" in st else vv.zero_()
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
                if "step" in st:
                    steps = max(steps, int(st["step"]))
                else:
                    st["step"] = torch.tensor(0.0)
        if self.mode != 2:
            # one shared step count drives the bias correction of the fused kernel
            self.steps = steps
            for p in self.flat.params:
                self.state[p]["step"] = torch.tensor(float(steps))
        self.step_dev.fill_(self.steps)
        self._lr_host = None
        self.sync_lr()


def set_deterministic(seed=42):
    """Seeds Python, numpy and torch and makes every step repeat bit for bit on the device (reference:
    utils/common/training_utils.py:27-35, whose cudnn.deterministic = True is ops.set_deterministic here)."""
    import random

    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
        ops.set_deterministic(True)


def get_optimizer(model, optimizer_name="adam", lr=0.001, weight_decay=1e-4):
    """Same factory signature as the reference (training_utils.py:311-331)."""
    return FusedOptimizer(model, optimizer_name, lr=lr, weight_decay=weight_decay)


def get_scheduler(optimizer, scheduler_name="step", step_size=20, gamma=0.1):
    """training_utils.py:333-353"""
    name = scheduler_name.lower()
    if name == "step":
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=step_size, gamma=gamma)
    if name == "cosine":
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=step_size)
    if name == "plateau":
        return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=gamma, patience=step_size)
    raise ValueError(f"Unsupported scheduler: {scheduler_name}")


# --------------------------------------------------------------------------------------
# bucketed gradient all-reduce
# --------------------------------------------------------------------------------------
class GradBucketer:
    """Splits the flat gradient buffer into contiguous buckets and all-reduces each bucket as soon as the
    backward pass has produced the gradients of all its parameters (signalled through ``notify``), on a
    side stream so that the collective overlaps the rest of backward.  Works on any backend
    (``nccl`` = RCCL on the GPUs, ``gloo`` on CPU for the tests)."""

    def __init__(self, flat, bucket_bytes=32 << 20, group=None, overlap=True):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # VC_DP_FORCE=1: issue the collectives even with one rank (a one-rank RCCL communicator exists), so that the
        # side-stream all-reduces, their fences and the phased-graph interleave run on the real backend on a one-GPU box
        self.collective = self.world > 1 or (dist.is_initialized() and os.environ.get("VC_DP_FORCE", "0") == "1")
        self.launches = 0  # collectives issued since construction (diagnostics: bench.py reports it per rank)
        self.on_gpu = flat.grad.is_cuda
        self.overlap = overlap and self.on_gpu
        self.stream = torch.cuda.Stream() if self.overlap else None
        # parameters are produced roughly in reverse registration order during backward, so buckets are
        # cut walking the flat buffer from its end
        self.buckets = []  # (start, end) element ranges
        self.bucket_of = {}
        limit = max(1, bucket_bytes // 4)
        end = flat.total
        cur_hi = end
        n = len(flat.params)
        members = []
        for i in range(n - 1, -1, -1):
            members.append(i)
            lo = flat.offsets[i]
            if cur_hi - lo >= limit or i == 0:
                b = len(self.buckets)
                self.buckets.append((lo, cur_hi))
                for j in members:
                    self.bucket_of[id(flat.params[j])] = b
                members = []
                cur_hi = lo
        self.sizes = [0] * len(self.buckets)
        for p in flat.params:
            self.sizes[self.bucket_of[id(p)]] += 1
        self.reset()

    def reset(self):
        self.pending = list(self.sizes)
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self.late = 0

    def notify(self, p):
        b = self.bucket_of.get(id(p))
        if b is None or self.launched[b]:
            return
        self.pending[b] -= 1
        if self.pending[b] <= 0:
            self._launch(b)

    def _launch(self, b):
        self.launched[b] = True
        if not self.collective:
            return
        self.launches += 1
        lo, hi = self.buckets[b]
        view = self.flat.grad[lo:hi]
        if self.on_gpu and dist.get_backend(self.group) == "gloo":
            # rehearsal of the multi-rank flow on a box without RCCL peers: reduce a host copy, synchronously
            host = view.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(host)
            return
        if self.overlap:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch whatever was not signalled, wait for every collective; gradients then hold the SUM over
        ranks (the optimizer folds the 1/world factor into its update)."""
        # `late` counts the buckets nobody launched during backward (their exchange is not overlapped): zero when every
        # kernel that writes a parameter gradient reports it; `total_late` accumulates over the run for diagnostics
        self.late = 0
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self.late += 1
                self._launch(b)
        self.total_late = getattr(self, "total_late", 0) + self.late
        for h in self.handles:
            h.wait()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.handles = []


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switches for a one-GPU box: every rank on device 0, collectives over gloo (RCCL refuses two
    # ranks on one device)
    if os.environ.get("VC_SINGLE_DEVICE", "0") == "1":
        local_rank = 0
    backend = backend or os.environ.get("VC_DIST_BACKEND") or None
    if (world > 1 or os.environ.get("VC_DP_FORCE", "0") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def make_data_parallel(model, optimizer, bucket_mb=32, group=None, overlap=True):
    """Broadcast rank 0's parameters and buffers, attach a GradBucketer to the optimizer and hook it
    into the kernels' gradient-ready notifications.  The model object is used as is (no wrapper, so
    state_dict keys carry no ``module.`` prefix)."""
    flat = optimizer.flat
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    forced = dist.is_initialized() and os.environ.get("VC_DP_FORCE", "0") == "1"
    if world > 1 or forced:
        dist.broadcast(flat.data, src=0, group=group)
        # buffers (BatchNorm running statistics, counters): one broadcast per dtype instead of hundreds of tiny ones
        by_dtype = {}
        for b in model.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            packed = torch.cat([b.detach().reshape(-1) for b in bufs])
            dist.broadcast(packed, src=0, group=group)
            o = 0
            for b in bufs:
                n = b.numel()
                b.copy_(packed[o:o + n].view_as(b))
                o += n
        ops.bump_weights_epoch()
    bucketer = GradBucketer(flat, bucket_bytes=bucket_mb << 20, group=group, overlap=overlap)
    optimizer.bucketer = bucketer
    optimizer.grad_scale = 1.0 / world
    ops.set_grad_ready_hook(bucketer.notify)
    ops.SIDE_STREAM_WORK = bucketer.collective and bucketer.overlap
    return bucketer


# --------------------------------------------------------------------------------------
# criterion, epoch loops, checkpoints
# --------------------------------------------------------------------------------------
class FusedCrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() replacement (mean reduction) on the fused HIP head; also records the arg-max
    predictions and the number of correct ones of the last call in ``self.stats``."""

    def __init__(self):
        super().__init__()
        self.stats = {}

    def forward(self, output, target):
        return ops.cross_entropy(output, target, self.stats)


def _run_batch(model, data, target, criterion):
    output = model(data)
    loss = criterion(output, target.squeeze())
    if isinstance(criterion, FusedCrossEntropyLoss):
        correct = criterion.stats["correct"]
    else:
        pred = output.argmax(dim=1, keepdim=True)
        correct = pred.eq(target.view_as(pred)).sum()
    return output, loss, correct


# VC_TRAIN_GRAPH=1: train_epoch replays the step as a hipGraph by default (its `graph` argument overrides)
TRAIN_GRAPH = os.environ.get("VC_TRAIN_GRAPH", "0") == "1"


def _graphed_batch(model, criterion, optimizer, data, target):
    """(loss, number of correct predictions) of one training step replayed as a hipGraph, or None when this batch has
    to run eagerly.  The first batch builds the graph: GraphedTrainStep runs exactly ONE eager step on it -- that
    batch's training step -- and captures; later batches of the same shape replay; batches of another shape (the
    short last batch of an epoch) run eagerly."""
    from .graph import GraphedTrainStep
    tgt = target.reshape(-1)
    # the graph is bound to the flat buffers of ONE optimizer and to ONE criterion's statistics: a call with another
    # of either builds a new graph instead of replaying into the old buffers
    key = (tuple(data.shape), data.dtype, tuple(tgt.shape), id(optimizer), id(criterion))
    cached = getattr(model, "_vc_train_graph", None)
    if cached is not None and cached[0][3:] != key[3:]:
        cached = None
    if cached is None:
        gs = GraphedTrainStep(model, optimizer, criterion, data, tgt, warmup=1)
        object.__setattr__(model, "_vc_train_graph", (key, gs))
        return gs.first_loss, gs.first_correct
    if cached[0] != key:
        return None
    loss = cached[1](data, tgt)
    return loss, cached[1].correct


def train_epoch(model, train_loader, criterion, optimizer, device, epoch, use_amp=False, scaler=None, logger=None,
                graph=None):
    """Returns (average loss over batches, accuracy %), as the reference's train_epoch.  Loss and accuracy
    are accumulated on the device and read back once per logging interval instead of three times per
    batch.  `use_amp`/`scaler` are accepted for signature compatibility; precision is chosen by the
    model's compute dtype.  graph=True replays the step (zero-grad, forward, loss, backward, optimizer) as one
    hipGraph for every batch of the first batch's shape instead of launching ~3000 kernels from Python; it needs the
    fused criterion and optimizer of this module."""
    model.train()
    use_graph = (TRAIN_GRAPH if graph is None else bool(graph)) and isinstance(criterion, FusedCrossEntropyLoss) \
        and isinstance(optimizer, FusedOptimizer)
    total_loss = torch.zeros((), device=device)
    correct = torch.zeros((), device=device, dtype=torch.int64)
    total = 0
    nb = 0
    for batch_idx, (data, target) in enumerate(train_loader):
        data, target = data.to(device, non_blocking=True), target.to(device, non_blocking=True)
        done = _graphed_batch(model, criterion, optimizer, data, target) if use_graph else None
        if done is not None:
            loss, ncorrect = done
        else:
            optimizer.zero_grad()
            _, loss, ncorrect = _run_batch(model, data, target, criterion)
            loss.backward()
            optimizer.step()
        total_loss += loss.detach()
        correct += ncorrect.reshape(()).to(torch.int64)
        total += target.size(0)
        nb += 1
        if logger and (batch_idx + 1) % 100 == 0:
            logger.info(f"Epoch {epoch} - Batch {batch_idx + 1}/{len(train_loader)} - Loss: {loss.item():.4f}, "
                        f"Acc: {100. * correct.item() / total:.2f}%")
            ops.check_lstm_error()  # the stream was just synchronised by .item(): the read costs nothing more
    result = total_loss.item() / max(nb, 1), 100. * correct.item() / max(total, 1)
    ops.check_lstm_error()
    return result


def validate_epoch(model, val_loader, criterion, device, epoch, use_amp=False, logger=None):
    model.eval()
    total_loss = torch.zeros((), device=device)
    correct = torch.zeros((), device=device, dtype=torch.int64)
    total = 0
    nb = 0
    with torch.no_grad():
        for batch_idx, (data, target) in enumerate(val_loader):
            data, target = data.to(device, non_blocking=True), target.to(device, non_blocking=True)
            _, loss, ncorrect = _run_batch(model, data, target, criterion)
            total_loss += loss
            correct += ncorrect.reshape(()).to(torch.int64)
            total += target.size(0)
            nb += 1
            if logger and (batch_idx + 1) % 50 == 0:
                logger.info(f"Epoch {epoch} - Val Batch {batch_idx + 1}/{len(val_loader)} - Loss: {loss.item():.4f}, "
                            f"Acc: {100. * correct.item() / total:.2f}%")
    result = total_loss.item() / max(nb, 1), 100. * correct.item() / max(total, 1)
    ops.check_lstm_error()
    return result


def save_checkpoint(model, optimizer, epoch, loss, save_path, scheduler=None):
    """Same dictionary as the reference (training_utils.py:212-219), written from rank 0 only.  With a scheduler
    its state goes in as one more key, ``scheduler_state_dict``; the reference's loader ignores keys it does not
    know, so the file stays loadable there."""
    if dist.is_initialized() and dist.get_rank() != 0:
        return
    os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
    ckpt = {"epoch": epoch, "model_state_dict": model.state_dict(),
            "optimizer_state_dict": optimizer.state_dict(), "loss": loss}
    if scheduler is not None:
        ckpt["scheduler_state_dict"] = scheduler.state_dict()
    torch.save(ckpt, save_path)


def strip_module_prefix(state_dict):
    """Checkpoints written under nn.DataParallel carry a ``module.`` prefix on every key."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def restore_scheduler(scheduler, ckpt, optimizer=None):
    """Puts a learning-rate scheduler where it stood when `ckpt` was written.  A checkpoint of this trainer
    carries the scheduler's own state.  One written by the reference does not (training_utils.py:212-217); its
    optimizer state holds the learning rate of that moment, so for the epoch-driven schedules (StepLR,
    CosineAnnealingLR: training_utils.py:345-348) setting the epoch counter is enough; ReduceLROnPlateau keeps the
    restored learning rate and starts its patience window afresh (its history was never saved)."""
    if scheduler is None:
        return
    if "scheduler_state_dict" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    elif not isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
        done = int(ckpt["epoch"])  # scheduler.step() ran once per finished epoch (train.py:668-683)
        scheduler.last_epoch = done
        scheduler._step_count = done + 1
        scheduler._last_lr = [g["lr"] for g in scheduler.optimizer.param_groups]
    opt = optimizer if optimizer is not None else getattr(scheduler, "optimizer", None)
    if isinstance(opt, FusedOptimizer):
        opt._lr_host = None
        opt.sync_lr()


def load_checkpoint(model, optimizer, checkpoint_path, scheduler=None):
    """Returns the epoch to start from (training_utils.py:222-243); accepts DataParallel-prefixed keys.  Unlike
    the reference's resume branch (train.py:848-873 only checks that the file exists) the states are restored:
    weights and buffers, optimizer moments and step count, the learning rate on the device, the scheduler, and the
    packed bf16 copies of the weights are marked stale so that the next forward rebuilds them."""
    if not os.path.exists(checkpoint_path):
        return 1
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    model.load_state_dict(strip_module_prefix(ckpt["model_state_dict"]))
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    restore_scheduler(scheduler, ckpt, optimizer)
    ops.bump_weights_epoch()
    return ckpt["epoch"] + 1


def checkpoint_path_for_epoch(ckpt_dir, epoch):
    """File name the reference's training loop gives the checkpoint of a finished epoch (train.py:699, 859)."""
    return os.path.join(ckpt_dir, f"checkpoint_epoch_{epoch}.pth")


def resume_training(model, optimizer, scheduler, ckpt_dir, resume_epoch):
    """``--resume N`` of the reference (train.py:779-780, 848-873) carried through: finds
    ``checkpoint_epoch_N.pth`` under `ckpt_dir`, restores model, optimizer and scheduler, and returns the epoch
    to continue with (N + 1).  Raises FileNotFoundError naming the checkpoints that do exist (the reference
    prints the same list and gives up)."""
    path = checkpoint_path_for_epoch(ckpt_dir, resume_epoch)
    if not os.path.exists(path):
        have = sorted(f for f in os.listdir(ckpt_dir) if f.startswith("checkpoint_epoch_")) \
            if os.path.isdir(ckpt_dir) else []
        raise FileNotFoundError(f"no checkpoint of epoch {resume_epoch} under {ckpt_dir}; available: {have}")
    if dist.is_initialized():
        # every rank reads the same file, so replicas restart identical without a broadcast
        dist.barrier()
    return load_checkpoint(model, optimizer, path, scheduler)


# --------------------------------------------------------------------------------------
# inference callers (forward only, eval-mode BatchNorm)
# --------------------------------------------------------------------------------------
def load_split_checkpoint(ckpt, encoder, decoder):
    """The per-script checkpoint layouts of the reference (predict.py:108-123; SwinCRNN.py:329-336;
    clusterSwin_TCGlstm.py:345-352): {cnn|swin|cluster_swin}_encoder_state_dict + {rnn|tcglstm}_decoder_state_dict.
    Returns True if `ckpt` had one of those layouts and both halves were loaded."""
    for ek, dk in (("cnn_encoder_state_dict", "rnn_decoder_state_dict"),
                   ("swin_encoder_state_dict", "rnn_decoder_state_dict"),
                   ("cluster_swin_encoder_state_dict", "tcglstm_decoder_state_dict")):
        if isinstance(ckpt, dict) and ek in ckpt and dk in ckpt:
            encoder.load_state_dict(strip_module_prefix(ckpt[ek]))
            decoder.load_state_dict(strip_module_prefix(ckpt[dk]))
            ops.bump_weights_epoch()
            return True
    return False


def predict_topk(model, clip, k=5):
    """Top-k class probabilities of ONE clip [1,T,3,H,W] (or [1,1,T,H,W] for Conv3D), as predict.py:134-142:
    eval mode, softmax over the logits, classes by descending probability.  Returns (probs[k], indices[k])."""
    model.eval()
    with torch.no_grad():
        logits = model(clip)
        probs = torch.softmax(logits.float(), dim=1)[0]
    k = min(k, probs.numel())
    p, idx = torch.sort(probs, descending=True, stable=True)
    p, idx = p[:k].cpu(), idx[:k].cpu()
    ops.check_lstm_error()
    return p, idx


def save_predictions(model, test_loader, device, save_path, action_names=None):
    """The reference's save_predictions (training_utils.py:245-281): eval-mode arg-max over a loader, written as
    a pickled DataFrame {'y', 'y_pred'} and returned.  Predictions stay on the device until the loader is
    exhausted (one read-back instead of two per batch)."""
    import pickle

    import pandas as pd
    model.eval()
    preds, labels = [], []
    with torch.no_grad():
        for data, target in test_loader:
            data = data.to(device, non_blocking=True)
            output = model(data)
            preds.append(output.argmax(dim=1))
            labels.append(target.reshape(-1))
    y_pred = torch.cat(preds).cpu().numpy().flatten() if preds else []
    y = torch.cat(labels).cpu().numpy().flatten() if labels else []
    ops.check_lstm_error()
    results_df = pd.DataFrame({"y": y, "y_pred": y_pred})
    if not (dist.is_initialized() and dist.get_rank() != 0):
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        with open(save_path, "wb") as f:
            pickle.dump(results_df, f)
    return results_df
