"""hipGraph capture of the whole training step.

A training step of these models is 1.5-3 thousand kernel launches; issued one by one from Python the host,
not the GPU, sets the step time for the smaller workloads (Swin at 8 clips: 33 ms of kernels in a 61 ms step).
``GraphedTrainStep`` records zero-grad + forward + loss + backward (+ the fused optimizer when there is a single
rank) once into a graph over static input buffers and replays it per batch.  Everything that changes from step
to step is read from device memory by the kernels (optimizer step count and learning rate, dropout seed
counter), so the replayed graph stays correct.

With more than one rank the gradient all-reduce and the optimizer stay outside the graph: the buckets are
reduced right after the replay (102 MB of fp32 gradients for ResNet-50 is ~2 ms on xGMI next to a ~75 ms step),
then the optimizer kernel runs.
"""
import torch
import torch.distributed as dist

from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, criterion, x_example, y_example, warmup=3):
        self.model, self.opt, self.crit = model, optimizer, criterion
        self.world = dist.get_world_size() if dist.is_initialized() else 1
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
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ops.set_grad_ready_hook(None)  # no collectives inside the capture
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.opt.flat.zero_grad()
                out = model(self.x)
                self.loss = criterion(out, self.y)
                self.loss.backward()
                if self.world == 1:
                    self.opt.launch()
            self.out = out
        finally:
            ops.set_grad_ready_hook(hook)
        ops.bump_weights_epoch()

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
        self.graph.replay()
        if self.world > 1:
            b = self.opt.bucketer
            if b is not None:
                b.reset()
                b.finish()
            self.opt.launch()
        self.opt.after_step()
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
