"""hipGraph capture of the whole training step.

A training step of these models is 1.5-3 thousand kernel launches; issued one by one from Python the host,
not the GPU, sets the step time for the smaller workloads (Swin at 8 clips: 33 ms of kernels in a 61 ms step).
``GraphedTrainStep`` records zero-grad + forward + loss + backward (+ the fused optimizer when there is a single
rank) once over static input buffers and replays it per batch.  Everything that changes from step to step is read
from device memory by the kernels (optimizer step count and learning rate, dropout seed counter), so the replayed
graphs stay correct.

The backward pass is captured in PHASES, one hipGraph each, cut at the activations the encoders mark at their stage
boundaries (``ops.mark_cut``): phase 0 = zero-grad, forward, loss and the backward pass down to the last mark, phase k
= the backward pass between two marks, the last phase = the rest (and the optimizer when there is one rank).  With more
than one rank the gradient buckets whose parameters a phase has completed are all-reduced (RCCL, side stream) while the
next phase replays, so the exchange overlaps the backward pass as it does in the eager executor; the optimizer kernel
is launched after the last bucket has arrived.  Collectives themselves are issued from the host between replays, not
captured.
"""
import torch
import torch.distributed as dist

from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, criterion, x_example, y_example, warmup=3, phased=True):
        self.model, self.opt, self.crit = model, optimizer, criterion
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # gradients are exchanged between replays (more than one rank, or a forced one-rank exchange: see GradBucketer)
        self.exchange = optimizer.bucketer is not None and optimizer.bucketer.collective
        self.x = x_example.clone()
        self.y = y_example.clone()
        model.train()
        if ops.PROFILE is not None:
            raise RuntimeError("disable ops.PROFILE while capturing")
        hook = ops._grad_ready_hook
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # real training steps on the example batch
                self.first_loss = self._eager().detach()
                stats = getattr(criterion, "stats", None) or {}
                self.first_correct = stats["correct"].clone() if "correct" in stats else None
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # no collectives inside the capture: the hook only notes in which phase a parameter's gradient is complete
        phase_of = {}
        cur = [0]
        ops.set_grad_ready_hook(lambda p: phase_of.__setitem__(id(p), cur[0]))
        ops._cut_recorder = [] if phased else None
        self.graphs = []
        try:
            g0 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g0):
                self.opt.flat.zero_grad()
                out = model(self.x)
                self.loss = criterion(out, self.y)
                # the tensors the captured loss kernel writes: an eager call of the same criterion (validation, a
                # short last batch) rebinds criterion.stats to fresh tensors, so replays must hand out THESE
                stats = getattr(criterion, "stats", None) or {}
                self.correct, self.pred = stats.get("correct"), stats.get("pred")
                cuts, ops._cut_recorder = (ops._cut_recorder or []), None
                if cuts:
                    grad = torch.autograd.grad([self.loss], [cuts[-1]])[0]
                else:
                    self.loss.backward()
                    if not self.exchange:
                        self.opt.launch()
            self.graphs.append(g0)
            for k in range(len(cuts) - 1, -1, -1):
                cur[0] += 1
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, pool=g0.pool()):  # one memory pool: activations of phase 0 are freed here
                    if k > 0:
                        grad = torch.autograd.grad([cuts[k]], [cuts[k - 1]], [grad])[0]
                    else:
                        torch.autograd.backward([cuts[0]], [grad])
                        if not self.exchange:
                            self.opt.launch()
                self.graphs.append(gk)
            self.out = out
        finally:
            ops._cut_recorder = None
            ops.set_grad_ready_hook(hook)
        del cuts
        # buckets to reduce after each phase: a bucket is complete once its last parameter is (parameters that no
        # kernel reported -- none today -- count as complete at the very end)
        self.phase_buckets = [[] for _ in self.graphs]
        b = self.opt.bucketer
        if b is not None:
            last = len(self.graphs) - 1
            done = [0] * len(b.buckets)
            for p in self.opt.flat.params:
                i = b.bucket_of[id(p)]
                done[i] = max(done[i], phase_of.get(id(p), last))
            for i, ph in enumerate(done):
                self.phase_buckets[ph].append(i)
        ops.bump_weights_epoch()

    @property
    def graph(self):  # the single graph of an unphased step (kept for callers that replay it themselves)
        return self.graphs[0]

    def _eager(self):
        self.opt.zero_grad()
        loss = self.crit(self.model(self.x), self.y)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self, x, y):
        """One training step on (x, y); returns the (device-resident) loss of this step."""
        if x is not self.x:
            self.x.copy_(x, non_blocking=True)
        if y is not self.y:
            self.y.copy_(y, non_blocking=True)
        self.opt.sync_lr()
        b = self.opt.bucketer if self.exchange else None
        if b is not None:
            b.reset()
        for g, buckets in zip(self.graphs, self.phase_buckets):
            g.replay()
            if b is not None:
                for i in buckets:  # their all-reduce runs beside the next phase's replay
                    b._launch(i)
        if self.exchange:
            b.finish()
            self.opt.launch()
        self.opt.after_step()
        if self.correct is not None:  # the criterion reports what this replay computed
            self.crit.stats["correct"], self.crit.stats["pred"] = self.correct, self.pred
        return self.loss


class GraphedInference:
    """Eval-mode forward of a model replayed as one hipGraph: logits = infer(x) for clips of the captured shape
    (predict.py / save_predictions style callers with fixed-size batches).  The returned tensor is the graph's
    static output buffer: copy it if it must survive the next call."""

    def __init__(self, model, x_example, warmup=2):
        self.model = model
        model.eval()
        self.x = x_example.clone()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):  # weight packs and scratch buffers are created here, outside capture
                    model(self.x)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = model(self.x)

    def __call__(self, x):
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.out
