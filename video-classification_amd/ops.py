"""Autograd-aware wrappers over the C ABI of libvc_hip (hand-written gfx950 kernels).

Activations are channels-last ("CL") tensors in the compute dtype (bfloat16 by default, float32 for
the exact path): ``[N, D, H, W, C]`` for convolutions, ``[M, C]`` for linear layers, with C a multiple
of 8.  PyTorch only owns the memory and the autograd graph; every arithmetic step below is a call
into libvc_hip.  Parameter gradients are accumulated by the kernels directly into ``param.grad``
(fp32), so the autograd functions return ``None`` for parameter inputs.
"""
import os

import torch

from . import _lib as L

# bumped whenever parameters are rewritten behind autograd's back (fused optimizer, load_state_dict)
_weights_epoch = 0


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


# When set to a list, every contraction launch appends (tag, algorithmic_flops, start_event, end_event),
# with the events recorded on the launch stream around the kernel (bench.py's roofline measurement).
PROFILE = None


def _timed(tag, flops, launch, nbytes=0.0):
    """Run `launch`; when PROFILE is a list also record (tag, algorithmic flops, HIP events, algorithmic bytes)."""
    if PROFILE is None:
        launch()
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    launch()
    e.record()
    PROFILE.append((tag, flops, s, e, nbytes))


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: libvc_hip kernels need tensors on the MI355X (got device {t.device}); "
                           "there is no CPU fallback")


def pad8(c):
    return (c + 7) // 8 * 8


def _grad_buf(p):
    """fp32 gradient buffer of a parameter (created zeroed on first use)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


_det_workspaces = {}


def set_deterministic(on=True, workspace_mb=256):
    """Bit-for-bit repeatable steps (the counterpart of torch.backends.cudnn.deterministic = True): sums that the kernels
    otherwise combine with fp32 atomics are added in a fixed order, through a workspace of `workspace_mb` MiB that is
    registered with the library for the current device.  Call once per device a model runs on.  About 5 % slower."""
    global WGRAD_STREAM
    if not on:
        L.call("vc_set_deterministic", 0, L.ptr(None), L.i64(0))
        _det_workspaces.clear()
        return
    dev = torch.device("cuda", torch.cuda.current_device())
    ws = _det_workspaces.get(dev)
    if ws is None or ws.numel() != workspace_mb << 18:
        ws = _det_workspaces[dev] = torch.empty(workspace_mb << 18, device=dev, dtype=torch.float32)
    WGRAD_STREAM = False  # the workspace is shared by every call: one stream at a time
    L.call("vc_set_deterministic", 1, L.ptr(ws), L.i64(ws.numel()))


def is_deterministic():
    return bool(L.query("vc_get_deterministic"))


_grad_ready_hook = None


def set_grad_ready_hook(fn):
    """fn(param) is called right after the kernels that produce param's gradient have been enqueued
    (the data-parallel trainer uses it to start a bucket's all-reduce while backward continues)."""
    global _grad_ready_hook
    _grad_ready_hook = fn


def _ready(*params):
    if _grad_ready_hook is not None:
        for p in params:
            if p is not None and p.requires_grad:
                _grad_ready_hook(p)


# While a list, every activation a model passes to mark_cut() during forward is appended to it (in forward order).
_cut_recorder = None


def mark_cut(t):
    """Marks `t` (an activation at a stage boundary of an encoder: everything computed after it depends on it) as a
    place where the backward pass may be cut into phases.  graph.GraphedTrainStep captures one hipGraph per phase, so
    that the gradient buckets a phase completed can be all-reduced while the next phase replays.  Returns `t` itself."""
    if _cut_recorder is not None and t.requires_grad and torch.is_grad_enabled():
        _cut_recorder.append(t)
    return t


def cast(t, dtype):
    """dtype conversion with the library's cast kernel (float32 <-> bfloat16)."""
    if t.dtype == dtype:
        return t
    t = t.contiguous()
    out = torch.empty_like(t, dtype=dtype)
    L.call("vc_cast", L.dtype_code(t.dtype), L.dtype_code(dtype), L.ptr(t), L.ptr(out), L.i64(t.numel()), L.stream())
    return out


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, dtype):
        ctx.src = t.dtype
        return cast(t, dtype)

    @staticmethod
    def backward(ctx, g):
        return cast(g, ctx.src), None


def cast_ag(t, dtype):
    return t if t.dtype == dtype else _Cast.apply(t, dtype)


def elementwise(op, a, b=None):
    a = a.contiguous()
    out = torch.empty_like(a)
    L.call("vc_elementwise", L.dtype_code(a.dtype), op, L.ptr(a), L.ptr(b.contiguous() if b is not None else None),
           L.ptr(out), L.i64(a.numel()), L.stream())
    return out


# ----------------------------------------------------------------------------------------------
# clip layout
# ----------------------------------------------------------------------------------------------
def clip_to_frames(x, dtype, cpad=8):
    """fp32 clip [B,T,C,H,W] -> channels-last frames [T*B, 1, H, W, cpad], time-major (frame = t*B + b)."""
    _require_cuda(x, "clip_to_frames")
    B, T, C, H, W = x.shape
    x = x.contiguous().float()
    out = torch.empty(T * B, 1, H, W, cpad, device=x.device, dtype=dtype)
    L.call("vc_clip_to_channels_last", L.dtype_code(dtype), L.ptr(x), L.ptr(out), B, T, C, H * W, cpad, 1, 0,
           L.stream())
    return out


def volume_to_cl(x, dtype, cpad=8):
    """fp32 volume [B,C,T,H,W] -> channels-last [B, T, H, W, cpad]."""
    _require_cuda(x, "volume_to_cl")
    B, C, T, H, W = x.shape
    x = x.contiguous().float()
    out = torch.empty(B, T, H, W, cpad, device=x.device, dtype=dtype)
    L.call("vc_clip_to_channels_last", L.dtype_code(dtype), L.ptr(x), L.ptr(out), B, T, C, H * W, cpad, 0, 1,
           L.stream())
    return out


# ----------------------------------------------------------------------------------------------
# convolution / linear
# ----------------------------------------------------------------------------------------------
def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def _packed_weights(weight, ci, taps, cip, cop, dtype, want_t=True):
    """(forward pack [cop][taps][cip], transposed pack [Ci][taps][cop]) in the compute dtype; want_t=False: the
    second is None (a layer that brings its own data-gradient pack)."""
    # The pack is cached on the parameter object itself (it dies with it) and is valid while neither
    # autograd's version counter nor the library's own "weights rewritten" epoch has moved.
    key = (weight.data_ptr(), weight._version, _weights_epoch, dtype, cip, cop, taps, want_t)
    cache = getattr(weight, "_vc_pack", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    co = weight.shape[0]
    # the pack kernel writes every element of the first co rows; only padded output rows need the zero fill
    wf = (torch.empty if cop == co else torch.zeros)(cop, taps, cip, device=weight.device, dtype=dtype)
    wt = torch.empty(ci, taps, cop, device=weight.device, dtype=dtype) if want_t else None
    w = weight.detach().contiguous()
    L.call("vc_pack_weight", L.dtype_code(dtype), L.ptr(w), L.ptr(wf), L.ptr(wt), co, ci, taps, cip, cop, L.stream())
    weight._vc_pack = (key, wf, wt)
    return wf, wt


def _wgrad_scratch(weight, shape):
    """fp32 accumulator of a packed weight gradient, owned by the parameter and kept ZERO between uses:
    vc_unpack_wgrad clears it while it reads it, so no per-step fill pass is launched."""
    buf = getattr(weight, "_vc_dwp", None)
    if buf is None or tuple(buf.shape) != tuple(shape) or buf.device != weight.device:
        buf = torch.zeros(shape, device=weight.device, dtype=torch.float32)
        weight._vc_dwp = buf
    return buf


def _zero_scratch(owner, name, shape, device):
    """Zero-initialised fp32 scratch cached on `owner` (a module); the kernels that consume it clear it again."""
    buf = getattr(owner, name, None) if owner is not None else None
    if buf is None or tuple(buf.shape) != tuple(shape) or buf.device != device:
        buf = torch.zeros(shape, device=device, dtype=torch.float32)
        if owner is not None:
            object.__setattr__(owner, name, buf)
    return buf


class _Conv(torch.autograd.Function):
    """y = conv(x, weight) + bias (optionally ReLU) on channels-last tensors."""

    @staticmethod
    def forward(ctx, x, weight, bias, kernel, stride, padding, relu, out_f32, ci, join, stats_groups, residual=None,
                pre_gelu=False, want_gelu=False, gelu_x=None, drop=None):
        _require_cuda(x, "conv")
        ctx.join = join
        ctx.bn_pre = getattr(x, "_vc_bn_pre", None) if (join is None and FUSE_BN_PRE) else None
        # drop = (p, samples): stochastic depth on this layer's result before the residual joins it; the rows split
        # into `samples` equal contiguous samples.  Kept as (p, seed, samples): backward scales the gradient alike.
        ctx.drop = None
        drop_done = False
        if drop is not None:
            import ctypes
            if residual is None:
                raise RuntimeError("conv: stochastic depth is applied to a residual branch (residual missing)")
            _dropout_seed[0] = (_dropout_seed[0] * 6364136223846793005 + 1442695040888963407) & 0x7FFFFFFFFFFFFFFF
            ctx.drop = (float(drop[0]), ctypes.c_uint64(_dropout_seed[0]), int(drop[1]))
        # the slab-sum output never carries a gradient: do not let autograd fill a zero tensor of its size per call
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        # pre_gelu: the layer's input is gelu(x); x itself (the GELU's input) is what autograd differentiates, so
        # that the data-gradient GEMM can apply gelu' in its epilogue instead of a separate pass
        ctx.pre = None
        if pre_gelu:
            ctx.pre = x
            # gelu(x) may come with x from the layer that produced it (want_gelu there), else it is a pass here
            x = gelu_x.contiguous().view(x.shape) if gelu_x is not None and gelu_x.numel() == x.numel() \
                else elementwise(3, x)
        dtype = x.dtype
        N, D, H, W, cip = x.shape
        co = weight.shape[0]
        ci = ci or weight.shape[1]
        if cip < ci or cip % 8:
            raise RuntimeError(f"conv: input has {cip} channels, weight expects {ci} (padded to a multiple of 8)")
        k, s, p = _triple(kernel), _triple(stride), _triple(padding)
        taps = k[0] * k[1] * k[2]
        cop = pad8(co)
        Do, Ho, Wo = [(i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip((D, H, W), k, s, p)]
        desc = [N, D, H, W, cip, Do, Ho, Wo, cop, *k, *s, *p]
        if weight.numel() != co * ci * taps:
            raise RuntimeError(f"conv: weight of shape {tuple(weight.shape)} is not [Co={co}, Ci={ci}, taps={taps}]")
        flat = taps > 1 and (Do, Ho, Wo) == (1, 1, 1) and p == (0, 0, 0) and k == (D, H, W)
        wf, wt = _packed_weights(weight, ci, taps, cip, cop, dtype, want_t=not flat)
        if flat:
            # the kernel covers the whole input: a Linear layer over the flattened channels-last map.  Same
            # forward pack; the data gradient uses a tap-major transposed pack (every tap is valid for exactly
            # one input pixel, so the gather form would waste taps-1 of every taps products).
            desc = [N, 1, 1, 1, taps * cip, 1, 1, 1, cop, 1, 1, 1, 1, 1, 1, 0, 0, 0]
            cache = getattr(weight, "_vc_pack_flat", None)
            if cache is None or cache[0] is not wf:
                wt = torch.empty(taps * cip, cop, device=x.device, dtype=dtype)
                L.call("vc_pack_weight_flat_t", L.dtype_code(dtype), L.ptr(weight.detach()), L.ptr(wt), co, ci, taps,
                       cip, cop, L.stream())
                weight._vc_pack_flat = (wf, wt)
            else:
                wt = cache[1]
        b = None
        if bias is not None:
            b = bias.detach()
            if cop != co:
                b = torch.cat([b, b.new_zeros(cop - co)])
        f32 = bool(out_f32) or dtype == torch.float32
        y = torch.empty(N, Do, Ho, Wo, cop, device=x.device, dtype=torch.float32 if f32 else dtype)
        flops = 2.0 * N * Do * Ho * Wo * co * ci * taps  # algorithmic: true channel counts, no padding
        M = N * Do * Ho * Wo
        partial = None
        g_out = None
        # algorithmic HBM bytes of one launch: the input, the packed weight and the output, each moved once
        nbytes = float(x.numel() * x.element_size() + wf.numel() * wf.element_size() + y.numel() * y.element_size())
        if (stats_groups and FUSE_BN_STATS and not f32 and M % stats_groups == 0 and (M // stats_groups) % 64 == 0 and
                L.query("vc_conv_fwd_stats_supported", L.int_array(desc), L.dtype_code(dtype), 0, 0) == 1):
            # the BatchNorm that follows takes its statistics from 64-row slab sums written by the epilogue
            partial = torch.empty(M // 64, 2, cop, device=x.device, dtype=torch.float32)
            _timed("conv_fwd", flops, lambda: L.call(
                "vc_conv_fwd_stats", L.int_array(desc), L.dtype_code(dtype), L.ptr(x), 0, L.ptr(wf), L.ptr(b),
                L.ptr(y), 0, int(relu), L.ptr(partial), L.stream()), nbytes)
        elif (want_gelu and residual is None and FUSE_EPILOGUE_SIDE and not f32 and not relu and
              L.query("vc_conv_epilogue_side_supported", L.int_array(desc), L.dtype_code(dtype), 0) == 1):
            g_out = torch.empty_like(y)  # gelu(y), written by the same epilogue
            _timed("conv_fwd", flops, lambda: L.call(
                "vc_conv_fwd_gelu_dual", L.int_array(desc), L.dtype_code(dtype), L.ptr(x), L.ptr(wf), L.ptr(b),
                L.ptr(y), L.ptr(g_out), L.stream()), nbytes + float(y.numel() * y.element_size()))
        elif (residual is not None and FUSE_EPILOGUE_SIDE and not f32 and not relu and
              tuple(residual.shape) == tuple(y.shape) and residual.dtype == y.dtype and
              L.query("vc_conv_epilogue_side_supported", L.int_array(desc), L.dtype_code(dtype), 0) == 1):
            res = residual.contiguous()
            scales, rps = None, 0
            if ctx.drop is not None:  # per-sample factors of the stochastic depth, applied to the rows in the epilogue
                rps = M // ctx.drop[2]
                scales = torch.empty(ctx.drop[2], device=x.device, dtype=torch.float32)
                L.call("vc_drop_path_scales", L.ptr(scales), ctx.drop[2], ctx.drop[0], ctx.drop[1],
                       L.ptr(seed_counter(x.device)), L.stream())
            _timed("conv_fwd", flops, lambda: L.call(
                "vc_conv_fwd_residual", L.int_array(desc), L.dtype_code(dtype), L.ptr(x), L.ptr(wf), L.ptr(b),
                L.ptr(res), L.ptr(scales), rps, L.ptr(y), L.stream()), nbytes + float(res.numel() * res.element_size()))
            residual = None  # added
            drop_done = True
        elif SPLIT_K and L.query("vc_conv_fwd_splitk_plan", L.int_array(desc), L.dtype_code(dtype)) > 1:
            # few output tiles, long reduction: (tile, slice) work units fill the chip, a second pass adds the slices
            ns = L.query("vc_conv_fwd_splitk_plan", L.int_array(desc), L.dtype_code(dtype))
            ws = torch.empty(ns, M, cop, device=x.device, dtype=torch.float32)
            _timed("conv_fwd", flops, lambda: L.call(
                "vc_conv_fwd_splitk", L.int_array(desc), L.dtype_code(dtype), L.ptr(x), L.ptr(wf), L.ptr(b), L.ptr(y),
                int(relu), int(f32), L.ptr(ws), L.stream()), nbytes)
        else:
            _timed("conv_fwd", flops, lambda: L.call(
                "vc_conv_fwd", L.int_array(desc), L.dtype_code(dtype), L.ptr(x), 0, L.ptr(wf), L.ptr(b), L.ptr(y), 0,
                int(relu), int(f32), 0, L.stream()), nbytes)
        if ctx.drop is not None and not drop_done:  # outside the fused epilogue: a separate scaling pass
            ys = torch.empty_like(y)
            L.call("vc_drop_path", L.dtype_code(y.dtype), L.ptr(y), L.ptr(ys), L.i64(y.numel()),
                   L.i64(y.numel() // ctx.drop[2]), ctx.drop[0], ctx.drop[1], L.ptr(seed_counter(x.device)), L.stream())
            y = ys
        if residual is not None:  # geometry or dtype outside the fused epilogue: a separate add pass
            y = elementwise(0, y, cast(residual.contiguous().view(y.shape), y.dtype))
        ctx.flops = flops
        ctx.desc, ctx.taps, ctx.relu, ctx.has_bias = desc, taps, relu, bias is not None
        ctx.dims = (co, ci, cip, cop)
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, wt, y if relu else None)
        if partial is None:
            partial = y.new_empty(0, dtype=torch.float32)
        if g_out is None:
            g_out = y.new_empty(0)
        ctx.mark_non_differentiable(partial, g_out)
        return y, partial, g_out

    @staticmethod
    def backward(ctx, dy, _dpartial=None, _dgelu=None):
        x, wt, y = ctx.saved_tensors
        co, ci, cip, cop = ctx.dims
        if dy is None:  # nothing flows back into this layer
            return (None,) * len(ctx.needs_input_grad)
        dtype = x.dtype
        code = L.dtype_code(dtype)
        dy = cast(dy.contiguous(), dtype)
        dres = dy if (len(ctx.needs_input_grad) > 11 and ctx.needs_input_grad[11]) else None
        if ctx.drop is not None:  # the branch's gradient carries the sample factors, the residual's does not
            dys = torch.empty_like(dy)
            L.call("vc_drop_path", code, L.ptr(dy), L.ptr(dys), L.i64(dy.numel()), L.i64(dy.numel() // ctx.drop[2]),
                   ctx.drop[0], ctx.drop[1], L.ptr(seed_counter(dy.device)), L.stream())
            dy = dys
        if ctx.relu:
            dy = elementwise(1, dy, cast(y, dtype))
        desc = L.int_array(ctx.desc)
        weight, bias = ctx.weight, ctx.bias
        # the bias gradient (column sums of dy) rides along with the weight-gradient pass over dy
        want_db = bias is not None and bias.requires_grad
        dbp = None
        if want_db:
            dbp = _grad_buf(bias) if cop == co else _zero_scratch(None, "", (cop,), x.device)
        # The weight gradient and the data gradient of a layer are independent: optionally (WGRAD_STREAM) the
        # first goes to a side stream; the main stream joins before anything can reuse the tensors they read.
        need_dx = ctx.needs_input_grad[0]
        side = _side_stream(x.device) if (WGRAD_STREAM and PROFILE is None and weight.requires_grad and need_dx) else None
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)

        def weight_grads():
            if weight.requires_grad:
                if ctx.taps == 1 and cip == ci and cop == co:
                    # the packed layout IS the parameter's layout: accumulate straight into the gradient
                    dwp = _grad_buf(weight)
                    _timed("conv_wgrad", ctx.flops, lambda: L.call(
                        "vc_conv_wgrad", desc, code, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dwp), L.ptr(dbp), 0, L.stream()))
                else:
                    dwp = _wgrad_scratch(weight, (cop, ctx.taps, cip))
                    _timed("conv_wgrad", ctx.flops, lambda: L.call(
                        "vc_conv_wgrad", desc, code, L.ptr(x), 0, L.ptr(dy), 0, L.ptr(dwp), L.ptr(dbp), 0, L.stream()))
                    L.call("vc_unpack_wgrad", L.ptr(dwp), L.ptr(_grad_buf(weight)), co, ci, ctx.taps, cip, 3,
                           L.stream())
            elif want_db:
                M = dy.numel() // cop
                L.call("vc_colsum", code, L.ptr(dy), L.ptr(dbp), M, cop, cop, L.stream())
            if want_db and cop != co:  # padded channel count: fold the first co sums into the parameter's gradient
                _grad_buf(bias).add_(dbp[:co])

        if side is not None:
            with torch.cuda.stream(side):
                weight_grads()
        else:
            weight_grads()
        dx = None
        if need_dx:
            if cip != ci:
                raise RuntimeError("conv: data gradient requested for a channel-padded input")
            join = ctx.join
            acc = join is not None and join.buf is not None
            dx = join.buf if acc else torch.empty_like(x)
            tail = join.tail if (join is not None and join.pending == 1 and FUSE_BN_TAIL) else None
            pre = ctx.bn_pre
            gelu_done = False
            if (ctx.pre is not None and not acc and FUSE_EPILOGUE_SIDE and
                    L.query("vc_conv_epilogue_side_supported", desc, code, 1) == 1):
                # dx is the gradient of the GELU's input: gelu' is applied in the GEMM's epilogue
                _timed("conv_dgrad", ctx.flops, lambda: L.call(
                    "vc_conv_dgrad_gelu", desc, code, L.ptr(dy), L.ptr(wt), L.ptr(ctx.pre), L.ptr(dx), L.stream()))
                gelu_done = True
            elif (tail is not None and tail.bits is not None and tuple(tail.y.shape) == tuple(dx.shape) and
                    L.query("vc_conv_dgrad_bn_tail_supported", desc, code) == 1):
                # x is the output of a residual block and this is the last contribution to its gradient: the
                # epilogue applies that block's ReLU decisions and writes the slab sums its BatchNorm backward needs
                M = dx.numel() // dx.shape[-1]
                partial = torch.empty((M + 63) // 64, 2, dx.shape[-1], device=dx.device, dtype=torch.float32)
                _timed("conv_dgrad", ctx.flops, lambda: L.call(
                    "vc_conv_dgrad_bn_tail", desc, code, L.ptr(dy), L.ptr(wt), L.ptr(dx), int(acc), L.ptr(tail.y),
                    L.ptr(tail.bits), L.ptr(partial), L.stream()))
                tail.partial = partial
            elif (pre is not None and not acc and ctx.pre is None and tuple(pre.y.shape) == tuple(dx.shape) and
                    pre.y.dtype == dx.dtype and
                    L.query("vc_conv_dgrad_bn_pre_supported", desc, code, pre.rpg) == 1):
                # x = relu(bn(y)) feeds this layer alone: the epilogue applies the ReLU decisions and writes the slab
                # sums of that BatchNorm's backward reductions
                M = dx.numel() // dx.shape[-1]
                partial = torch.empty(M // 64, 2, dx.shape[-1], device=dx.device, dtype=torch.float32)
                _timed("conv_dgrad", ctx.flops, lambda: L.call(
                    "vc_conv_dgrad_bn_pre", desc, code, L.ptr(dy), L.ptr(wt), L.ptr(dx), L.ptr(pre.y), L.ptr(pre.stat),
                    pre.rpg, L.ptr(partial), L.stream()))
                pre.partial = partial
            else:
                _timed("conv_dgrad", ctx.flops, lambda: L.call(
                    "vc_conv_dgrad", desc, code, L.ptr(dy), 0, L.ptr(wt), L.ptr(dx), 0, int(dtype == torch.float32),
                    int(acc), L.stream()))
            if ctx.pre is not None and not gelu_done:  # unfused: the gradient w.r.t. gelu(x) times gelu'(x)
                dx = elementwise(4, dx, ctx.pre)
            if join is not None:
                dx = join.arrive(dx)
        if side is not None:
            main.wait_stream(side)
        _ready(weight, bias)
        grads = (dx, None, None, None, None, None, None, None, None, None, None, dres, None, None, None, None)
        return grads[:len(ctx.needs_input_grad)]


# VC_WGRAD_STREAM=1 runs a layer's weight gradient on a side stream beside its data gradient.  Measured on the
# bench workloads it LOSES 3-6 % (both kernels are sized to fill the chip: sharing it costs L2 locality and
# gains no idle time back), so the default keeps one stream.
WGRAD_STREAM = os.environ.get("VC_WGRAD_STREAM", "0") == "1"
_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


# VC_FUSE_EPILOGUE=0 keeps residual adds and the GELU gradient in their own elementwise passes (A/B measurements)
FUSE_EPILOGUE_SIDE = os.environ.get("VC_FUSE_EPILOGUE", "1") != "0"
# VC_SPLIT_K=0 keeps every forward layer's reduction whole (A/B measurements)
SPLIT_K = os.environ.get("VC_SPLIT_K", "1") != "0"
# VC_FUSE_BN_STATS=0 keeps BatchNorm statistics in their own pass over y (A/B measurements)
FUSE_BN_STATS = os.environ.get("VC_FUSE_BN_STATS", "1") != "0"


class _InputConv(torch.autograd.Function):
    """First convolution of a network, straight from the raw fp32 clip (1 or 3 channels).

    The kw taps along W are folded into the channel axis by the layout kernel (vc_clip_unfold_w), so the layer
    is a (kd, kh, 1) convolution over pad8(kw*C) channels instead of a (kd, kh, kw) one over C padded to 8:
    2.3x fewer products for a 7x7 / 3-channel stem, 3x for a 3x3x3 / 1-channel one.  The clip needs no gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, kernel, stride, padding, layout, dtype, stats_groups):
        _require_cuda(x, "input_conv")
        ctx.set_materialize_grads(False)
        x = x.contiguous().float()
        k, s, p = _triple(kernel), _triple(stride), _triple(padding)
        if layout == "clip":      # [B,T,C,H,W] -> frames, time-major (frame = t*B + b), depth 1
            B, T, C, H, W = x.shape
            N, D, tm, chan_outer = T * B, 1, 1, 0
        elif layout == "volume":  # [B,C,T,H,W] -> [B,T,H,W]
            B, C, T, H, W = x.shape
            N, D, tm, chan_outer = B, T, 0, 1
        else:
            raise ValueError(layout)
        co, ci = weight.shape[0], weight.shape[1]
        if ci != C or weight.numel() != co * ci * k[0] * k[1] * k[2]:
            raise RuntimeError(f"input_conv: weight {tuple(weight.shape)} does not match {C} channels / kernel {k}")
        code = L.dtype_code(dtype)
        kw, sw, pw = k[2], s[2], p[2]
        Wo = (W + 2 * pw - kw) // sw + 1
        Jp = pad8(kw * C)
        U = torch.empty(N, D, H, Wo, Jp, device=x.device, dtype=dtype)
        L.call("vc_clip_unfold_w", code, L.ptr(x), L.ptr(U), B, T, C, H, W, kw, sw, pw, Jp, tm, chan_outer, L.stream())
        to = k[0] * k[1]
        cop = pad8(co)
        Do = (D + 2 * p[0] - k[0]) // s[0] + 1
        Ho = (H + 2 * p[1] - k[1]) // s[1] + 1
        desc = [N, D, H, Wo, Jp, Do, Ho, Wo, cop, k[0], k[1], 1, s[0], s[1], 1, p[0], p[1], 0]
        key = (weight.data_ptr(), weight._version, _weights_epoch, dtype, Jp, cop, to)
        cache = getattr(weight, "_vc_pack_u", None)
        if cache is not None and cache[0] == key:
            wf = cache[1]
        else:
            wf = torch.empty(cop, to, Jp, device=x.device, dtype=dtype)
            L.call("vc_pack_weight_unfold", code, L.ptr(weight.detach().contiguous()), L.ptr(wf), co, ci, to, kw, Jp,
                   cop, L.stream())
            weight._vc_pack_u = (key, wf)
        b = None
        if bias is not None:
            b = bias.detach()
            if cop != co:
                b = torch.cat([b, b.new_zeros(cop - co)])
        f32 = dtype == torch.float32
        y = torch.empty(N, Do, Ho, Wo, cop, device=x.device, dtype=dtype)
        M = N * Do * Ho * Wo
        flops = 2.0 * M * co * ci * to * kw
        nbytes = float(U.numel() * U.element_size() + wf.numel() * wf.element_size() + y.numel() * y.element_size())
        partial = None
        if (stats_groups and FUSE_BN_STATS and not f32 and M % stats_groups == 0 and (M // stats_groups) % 64 == 0 and
                L.query("vc_conv_fwd_stats_supported", L.int_array(desc), code, 0, 0) == 1):
            partial = torch.empty(M // 64, 2, cop, device=x.device, dtype=torch.float32)
            _timed("conv_fwd", flops, lambda: L.call(
                "vc_conv_fwd_stats", L.int_array(desc), code, L.ptr(U), 0, L.ptr(wf), L.ptr(b), L.ptr(y), 0, 0,
                L.ptr(partial), L.stream()), nbytes)
        else:
            _timed("conv_fwd", flops, lambda: L.call(
                "vc_conv_fwd", L.int_array(desc), code, L.ptr(U), 0, L.ptr(wf), L.ptr(b), L.ptr(y), 0, 0, int(f32), 0,
                L.stream()), nbytes)
        ctx.desc, ctx.flops, ctx.dims = desc, flops, (co, ci, to, kw, Jp, cop)
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(U)
        if partial is None:
            partial = y.new_empty(0, dtype=torch.float32)
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, dy, _dpartial=None):
        if dy is None:
            return (None,) * 9
        (U,) = ctx.saved_tensors
        co, ci, to, kw, Jp, cop = ctx.dims
        dtype = U.dtype
        code = L.dtype_code(dtype)
        dy = cast(dy.contiguous(), dtype)
        weight, bias = ctx.weight, ctx.bias
        want_db = bias is not None and bias.requires_grad
        dbp = None
        if want_db:
            dbp = _grad_buf(bias) if cop == co else _zero_scratch(None, "", (cop,), U.device)
        desc = L.int_array(ctx.desc)
        if weight.requires_grad:
            dwp = _wgrad_scratch(weight, (cop, to, Jp))
            _timed("conv_wgrad", ctx.flops, lambda: L.call(
                "vc_conv_wgrad", desc, code, L.ptr(U), 0, L.ptr(dy), 0, L.ptr(dwp), L.ptr(dbp), 0, L.stream()))
            L.call("vc_unpack_wgrad_unfold", L.ptr(dwp), L.ptr(_grad_buf(weight)), co, ci, to, kw, Jp, 3, L.stream())
        elif want_db:
            L.call("vc_colsum", code, L.ptr(dy), L.ptr(dbp), dy.numel() // cop, cop, cop, L.stream())
        if want_db and cop != co:
            _grad_buf(bias).add_(dbp[:co])
        _ready(weight, bias)
        return None, None, None, None, None, None, None, None, None


def input_conv(x, weight, bias, kernel, stride, padding, layout, dtype, bn=None, groups=1, relu=False, pool=None):
    """conv (+ BatchNorm + ReLU when `bn` is given) of the raw fp32 clip: layout "clip" = [B,T,C,H,W] evaluated
    per frame (time-major frames, 2-D kernel given as (1, kh, kw)), "volume" = [B,C,T,H,W] (3-D kernel)."""
    training = bn is not None and (bn.training or bn.running_mean is None)
    y, partial = _InputConv.apply(x, weight, bias, kernel, stride, padding, layout, dtype, groups if training else 0)
    if bn is None:
        return y
    if pool is not None:  # (k, s, p) of the max-pool that follows BatchNorm + ReLU
        return batch_norm_relu_pool(y, bn, groups, *pool, partial=partial if partial.numel() else None)
    return batch_norm(y, bn, groups, relu=relu, partial=partial if partial.numel() else None)


class BNTail:
    """What the producer of the LAST gradient contribution needs in order to finish the backward reductions of
    out = relu(bn(y) + shortcut) in its own epilogue: y, the recorded ReLU decisions, and (set by that producer)
    the slab sums it wrote.  Travels as the attribute ``_vc_bn_tail`` of `out`."""

    def __init__(self):
        self.y = self.bits = self.partial = None


class BNPre:
    """Travels as the attribute ``_vc_bn_pre`` of a = relu(bn(y)) when the caller states that ONE convolution consumes
    a: that convolution's data-gradient kernel then masks its result with the ReLU decisions (recomputed from y) and
    leaves the slab sums the BatchNorm backward needs (`partial`), so no reduction pass reads the gradient and y."""

    def __init__(self, y, stat, rpg):
        self.y, self.stat, self.rpg, self.partial = y, stat, rpg, None


# VC_FUSE_BN_PRE=0 keeps the BatchNorm backward reductions inside blocks in their own pass (A/B measurements)
FUSE_BN_PRE = os.environ.get("VC_FUSE_BN_PRE", "1") != "0"

# VC_FUSE_BN_TAIL=0 keeps the BatchNorm backward reductions of residual blocks in their own pass (A/B measurements)
FUSE_BN_TAIL = os.environ.get("VC_FUSE_BN_TAIL", "1") != "0"


class GradJoin:
    """Gradient accumulator of an activation that has several consumers (the input of a residual block).
    Each consumer's backward adds its contribution into ONE buffer inside its own kernel (the data-gradient
    GEMM accumulates in its epilogue) and only the last consumer to arrive hands the buffer to autograd, so
    no separate elementwise add pass runs over the activation."""

    def __init__(self, consumers, tail=None):
        self.consumers = self.pending = consumers
        self.buf = None
        self.tail = tail  # BNTail of the residual block whose output this activation is (or None)

    def arrive(self, buf):
        self.buf = buf
        self.pending -= 1
        if self.pending == 0:
            self.buf = None
            self.pending = self.consumers  # armed again: a retained graph may be walked a second time
            return buf
        return None


def conv(x, weight, bias=None, kernel=1, stride=1, padding=0, relu=False, out_f32=False, ci=None, join=None):
    """x: [N,D,H,W,C] channels-last.  weight: fp32 parameter whose memory is [Co][Ci][taps] (a Conv2d/Conv3d
    weight, or a Linear weight [Co, Ci*taps] read as a convolution over the whole feature map when `ci` is given).
    join: optional GradJoin shared with the other consumers of x."""
    return _Conv.apply(x, weight, bias, kernel, stride, padding, relu, out_f32, ci, join, 0)[0]


# VC_FOLD_EVAL_BN=0 keeps eval-mode BatchNorm as a pass of its own behind the convolution (A/B measurements)
FOLD_EVAL_BN = os.environ.get("VC_FOLD_EVAL_BN", "1") != "0"
_bn_stats_epoch = 0  # moves whenever a training-mode BatchNorm has rewritten running statistics through raw pointers


def _folded_bn_weights(weight, bias, bn):
    """(weight * s, beta + (bias - running_mean) * s) with s = gamma / sqrt(running_var + eps) per output channel: the
    convolution that equals BatchNorm(conv(x)) in eval mode.  fp32, cached on the BatchNorm module while neither the
    parameters nor the running statistics have moved."""
    key = (weight.data_ptr(), weight._version, _weights_epoch, _bn_stats_epoch, bn.running_mean._version,
           bn.running_var._version, None if bn.weight is None else bn.weight._version,
           None if bn.bias is None else bn.bias._version, None if bias is None else bias._version)
    cache = getattr(bn, "_vc_fold", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    with torch.no_grad():
        s = torch.rsqrt(bn.running_var.float() + bn.eps)
        if bn.weight is not None:
            s = s * bn.weight.float()
        wf = (weight.detach().float() * s.view(-1, *([1] * (weight.dim() - 1)))).contiguous()
        b0 = -bn.running_mean.float() if bias is None else bias.detach().float() - bn.running_mean.float()
        bf = b0 * s
        if bn.bias is not None:
            bf = bf + bn.bias.float()
    object.__setattr__(bn, "_vc_fold", (key, wf, bf.contiguous()))
    return wf, bf


def _conv_folded_bn(x, weight, bias, bn, kernel, stride, padding, relu, residual):
    """Eval-mode conv -> BatchNorm (-> + residual) (-> ReLU) as ONE convolution with folded weights: the activation is
    written once, by the GEMM's epilogue (inference only: no autograd graph is recorded)."""
    wf, bf = _folded_bn_weights(weight, bias, bn)
    if residual is None:
        return _Conv.apply(x, wf, bf, kernel, stride, padding, bool(relu), False, None, None, 0)[0]
    x = x.contiguous()
    N, D, H, W, cip = x.shape
    co, ci = wf.shape[0], wf.shape[1]
    k, st, p = _triple(kernel), _triple(stride), _triple(padding)
    taps = k[0] * k[1] * k[2]
    cop = pad8(co)
    Do, Ho, Wo = [(i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip((D, H, W), k, st, p)]
    desc = L.int_array([N, D, H, W, cip, Do, Ho, Wo, cop, *k, *st, *p])
    code = L.dtype_code(x.dtype)
    res = residual.contiguous()
    if (relu and cop == co and tuple(res.shape) == (N, Do, Ho, Wo, cop) and res.dtype == x.dtype and
            L.query("vc_conv_epilogue_side_supported", desc, code, 0) == 1):
        wp, _ = _packed_weights(wf, ci, taps, cip, cop, x.dtype, want_t=False)
        y = torch.empty(N, Do, Ho, Wo, cop, device=x.device, dtype=x.dtype)
        L.call("vc_conv_fwd_residual_relu", desc, code, L.ptr(x), L.ptr(wp), L.ptr(bf), L.ptr(res), L.ptr(y), L.stream())
        return y
    y = _Conv.apply(x, wf, bf, kernel, stride, padding, False, False, None, None, 0)[0]
    y = elementwise(0, y, cast(res.view(y.shape), y.dtype))
    return elementwise(2, y) if relu else y


def conv_bn(x, weight, bn, groups, bias=None, kernel=1, stride=1, padding=0, relu=False, residual=None, join=None,
            res_join=None, grad_inplace=False, pool=None, single_use=False):
    """BatchNorm(conv(x)) (+ residual, ReLU).  In training the convolution's epilogue also writes the slab sums
    the BatchNorm statistics are taken from, where the geometry allows it (otherwise a separate pass reads y)."""
    training = bn.training or bn.running_mean is None
    if (not training and FOLD_EVAL_BN and pool is None and join is None and not torch.is_grad_enabled() and
            x.dtype == torch.bfloat16):
        return _conv_folded_bn(x, weight, bias, bn, kernel, stride, padding, relu, residual)
    y, partial, _ = _Conv.apply(x, weight, bias, kernel, stride, padding, False, False, None, join,
                                groups if training else 0)
    if pool is not None:  # (k, s, p) of the max-pool that follows BatchNorm + ReLU
        return batch_norm_relu_pool(y, bn, groups, *pool, partial=partial if partial.numel() else None)
    return batch_norm(y, bn, groups, relu=relu, residual=residual, res_join=res_join,
                      partial=partial if partial.numel() else None, grad_inplace=grad_inplace, single_use=single_use)


def linear(x, weight, bias=None, relu=False, out_f32=False, residual=None, pre_gelu=False, want_gelu=False,
           gelu_x=None, drop_path=0.0, samples=0):
    """x: [M, C] -> [M, pad8(Co)].  residual: [M, pad8(Co)] added to the result (in the GEMM's epilogue where the
    geometry allows).  pre_gelu: the layer is applied to gelu(x); the gradient through the GELU is then taken in the
    data-gradient GEMM's epilogue; gelu_x, if given, is gelu(x) as the producing layer wrote it.  want_gelu: returns
    (y, gelu(y)), the second from the same epilogue (None where the geometry does not allow it: the consumer then
    computes it).  drop_path > 0 (with residual): stochastic depth, the result is residual + f * (x W^T + b) with one
    factor f in {0, 1 / (1 - drop_path)} per sample, the rows being `samples` equal contiguous samples."""
    M, C = x.shape
    drop = (drop_path, samples) if (drop_path > 0.0 and residual is not None) else None
    if residual is None and not pre_gelu and not want_gelu:
        y = conv(x.view(M, 1, 1, 1, C), weight, bias, 1, 1, 0, relu, out_f32)
        return y.view(M, y.shape[-1])
    res = residual.view(M, 1, 1, 1, residual.shape[-1]) if residual is not None else None
    y, _, g = _Conv.apply(x.view(M, 1, 1, 1, C), weight, bias, 1, 1, 0, relu, out_f32, None, None, 0, res, pre_gelu,
                          want_gelu, gelu_x, drop)
    y = y.view(M, y.shape[-1])
    if want_gelu:
        return y, (g.view(M, y.shape[-1]) if g.numel() else None)
    return y


# ----------------------------------------------------------------------------------------------
# BatchNorm with per-group (per time step) statistics
# ----------------------------------------------------------------------------------------------
def _nbt(bn):
    """num_batches_tracked of a BatchNorm module when it lives on the GPU as int64 (it is then counted by the
    finalize kernel itself), else None (the caller adds on the host side)."""
    t = getattr(bn, "num_batches_tracked", None) if bn is not None else None
    return t if (t is not None and t.is_cuda and t.dtype == torch.int64) else None


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, groups, momentum, eps, relu, residual, training,
                res_join, owner, partial, grad_inplace=False, tail=None, pre=None):
        _require_cuda(y, "batch_norm")
        ctx.res_join = res_join
        ctx.grad_inplace = grad_inplace
        ctx.tail = tail
        ctx.bn_pre = None
        ctx.owner = owner
        y = y.contiguous()
        C = y.shape[-1]
        M = y.numel() // C
        code = L.dtype_code(y.dtype)
        dev = y.device
        ga = gamma.detach() if gamma is not None else None
        be = beta.detach() if beta is not None else None
        if training:
            G = groups
            if M % G:
                raise RuntimeError(f"batch_norm: {M} rows do not split into {G} groups")
            rpg = M // G
            if rpg == 1:  # torch's BatchNorm raises the same way (a single frame through the BatchNorm1d heads)
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(y.shape)}")
            global _bn_stats_epoch
            _bn_stats_epoch += 1  # running statistics are rewritten through raw pointers: folded weights go stale
            sums = _zero_scratch(owner, "_vc_sums", (2, G, C), dev)  # cleared again by vc_bn_finalize
            stat = torch.empty(4, G, C, device=dev, dtype=torch.float32)  # mean, invstd, scale, shift
            if partial is not None:
                L.call("vc_bn_stats_from_partial", L.ptr(partial), L.ptr(sums[0]), L.ptr(sums[1]), G, rpg // 64, C,
                       L.stream())
            else:
                L.call("vc_bn_stats", code, L.ptr(y), L.ptr(sums[0]), L.ptr(sums[1]), G, rpg, C, L.stream())
            upd = running_mean is not None
            L.call("vc_bn_finalize_count", L.ptr(sums[0]), L.ptr(sums[1]), L.ptr(ga), L.ptr(be), L.ptr(running_mean),
                   L.ptr(running_var), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(stat[2]), L.ptr(stat[3]), G, C, rpg,
                   float(eps), float(momentum), int(upd) | 2, L.ptr(_nbt(owner)), L.stream())
        else:
            G, rpg = 1, M
            stat = torch.empty(4, 1, C, device=dev, dtype=torch.float32)
            L.call("vc_bn_eval_params", L.ptr(ga), L.ptr(be), L.ptr(running_mean), L.ptr(running_var), L.ptr(stat[2]),
                   L.ptr(stat[3]), C, float(eps), L.stream())
        out = torch.empty_like(y)
        res = residual.contiguous() if residual is not None else None
        bits = None
        if grad_inplace and relu and res is not None and training and BN_INPLACE_MASK:
            # one byte per 8 channels: backward takes the ReLU decisions from it instead of re-reading `out`
            bits = torch.empty(M, C // 8, device=dev, dtype=torch.uint8)
            L.call("vc_bn_apply_mask", code, L.ptr(y), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(res), L.ptr(out),
                   L.ptr(bits), G, rpg, C, 1, L.stream())
            if tail is not None and rpg % 64 == 0 and y.dtype == torch.bfloat16:
                tail.y, tail.bits = y, bits
        else:
            L.call("vc_bn_apply", code, L.ptr(y), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(res), L.ptr(out), G, rpg, C,
                   int(relu), L.stream())
        if pre is not None and training and relu and res is None and y.dtype == torch.bfloat16 and rpg % 64 == 0:
            pre.y, pre.stat, pre.rpg = y, stat, rpg
            ctx.bn_pre = pre
        ctx.G, ctx.rpg, ctx.C, ctx.relu, ctx.training = G, rpg, C, relu, training
        ctx.has_res = residual is not None
        ctx.gamma, ctx.beta = gamma, beta
        # the ReLU mask is recomputed from y in backward unless a residual entered in front of the ReLU
        ctx.save_for_backward(y, out if (relu and residual is not None and bits is None) else None, stat, bits)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise RuntimeError("batch_norm: backward through eval-mode BatchNorm is not implemented")
        y, out, stat, bits = ctx.saved_tensors
        G, rpg, C = ctx.G, ctx.rpg, ctx.C
        code = L.dtype_code(y.dtype)
        dout = cast(dout.contiguous(), y.dtype)
        s = _zero_scratch(ctx.owner, "_vc_bsums", (2, G, C), y.device)  # cleared again by vc_bn_bwd_params
        dy = torch.empty_like(y)
        if bits is not None:
            # the caller owns dout: the reduce pass masks it in place, the masked gradient is the residual's
            # gradient as it stands and the apply pass reads it without the activation
            tail = ctx.tail
            if tail is not None and tail.partial is not None:
                # the kernel that completed dout has already masked it and left slab sums of dz and dz*y
                L.call("vc_bn_bwd_sums_from_partial", L.ptr(tail.partial), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(s[0]),
                       L.ptr(s[1]), G, rpg // 64, C, L.stream())
                tail.partial = None
            else:
                L.call("vc_bn_bwd_reduce_mask", code, L.ptr(y), L.ptr(dout), L.ptr(bits), L.ptr(stat[0]),
                       L.ptr(stat[1]), L.ptr(s[0]), L.ptr(s[1]), G, rpg, C, L.stream())
            dres = dout
            L.call("vc_bn_bwd_apply", code, L.ptr(y), L.ptr(dout), L.ptr(None), L.ptr(stat[0]), L.ptr(stat[1]),
                   L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(s[0]), L.ptr(s[1]), L.ptr(dy), L.ptr(None), G, rpg, C, 0,
                   L.stream())
        elif ctx.bn_pre is not None and ctx.bn_pre.partial is not None:
            # the data-gradient kernel that produced dout has masked it and left slab sums of dz and dz * y
            pre = ctx.bn_pre
            L.call("vc_bn_bwd_sums_from_partial", L.ptr(pre.partial), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(s[0]),
                   L.ptr(s[1]), G, rpg // 64, C, L.stream())
            pre.partial = None
            dres = None
            L.call("vc_bn_bwd_apply", code, L.ptr(y), L.ptr(dout), L.ptr(None), L.ptr(stat[0]), L.ptr(stat[1]),
                   L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(s[0]), L.ptr(s[1]), L.ptr(dy), L.ptr(None), G, rpg, C, 0,
                   L.stream())
        else:
            L.call("vc_bn_bwd_reduce", code, L.ptr(y), L.ptr(dout), L.ptr(out), L.ptr(stat[0]), L.ptr(stat[1]),
                   L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(s[0]), L.ptr(s[1]), G, rpg, C, int(ctx.relu), L.stream())
            dres = torch.empty_like(y) if ctx.has_res else None
            L.call("vc_bn_bwd_apply", code, L.ptr(y), L.ptr(dout), L.ptr(out), L.ptr(stat[0]), L.ptr(stat[1]),
                   L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(s[0]), L.ptr(s[1]), L.ptr(dy), L.ptr(dres), G, rpg, C,
                   int(ctx.relu), L.stream())
        join = ctx.res_join
        if join is not None and dres is not None:
            if join.buf is not None:
                dres = elementwise(0, join.buf, dres)
            dres = join.arrive(dres)
        dg = _grad_buf(ctx.gamma) if ctx.gamma is not None and ctx.gamma.requires_grad else None
        db = _grad_buf(ctx.beta) if ctx.beta is not None and ctx.beta.requires_grad else None
        L.call("vc_bn_bwd_params", L.ptr(s[0]), L.ptr(s[1]), L.ptr(dg), L.ptr(db), G, C, 3, L.stream())
        _ready(ctx.gamma, ctx.beta)
        return dy, None, None, None, None, None, None, None, None, dres, None, None, None, None, None, None, None


# VC_BN_INPLACE_MASK=0 keeps the gradient of a residual block's output untouched (A/B measurements)
BN_INPLACE_MASK = os.environ.get("VC_BN_INPLACE_MASK", "1") != "0"


def batch_norm(y, bn, groups, relu=False, residual=None, res_join=None, partial=None, grad_inplace=False,
               single_use=False):
    """bn: an nn.BatchNorm{1,2,3}d module holding the parameters/buffers (its own forward is not used).
    res_join: optional GradJoin shared with the other consumers of `residual`.
    grad_inplace: the caller guarantees that nothing else reads the gradient of this call's output, so the
    backward pass may overwrite it (relu(bn(y) + residual): the masked gradient replaces it in place).
    single_use: the caller guarantees that the result feeds exactly ONE ops.conv / ops.conv_bn call and nothing else;
    that layer's data-gradient kernel then also does this BatchNorm's backward reductions (see BNPre)."""
    training = bn.training or bn.running_mean is None
    tail = BNTail() if (grad_inplace and relu and residual is not None and training and torch.is_grad_enabled()) else None
    pre = BNPre(None, None, 0) if (single_use and relu and residual is None and training and FUSE_BN_PRE and
                                   torch.is_grad_enabled()) else None
    out = _BatchNorm.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, groups,
                           bn.momentum if bn.momentum is not None else 0.1, bn.eps, relu, residual, training,
                           res_join, bn, partial, grad_inplace, tail, pre)
    if tail is not None and tail.bits is not None:
        out._vc_bn_tail = tail
    if pre is not None and pre.y is not None:
        out._vc_bn_pre = pre
    if training and bn.num_batches_tracked is not None and _nbt(bn) is None:
        bn.num_batches_tracked += groups
    return out


class _BatchNormPool(torch.autograd.Function):
    """max_pool(relu(batch_norm(y))) per frame without the full-resolution activation in between: y [N,D,H,W,C]
    (N*D frames of H x W) -> [N,D,Ho,Wo,C]."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, groups, momentum, eps, training, owner, partial,
                k, s, p):
        _require_cuda(y, "batch_norm_pool")
        y = y.contiguous()
        N, D, H, W, C = y.shape
        F_ = N * D
        M = F_ * H * W
        code = L.dtype_code(y.dtype)
        dev = y.device
        ga = gamma.detach() if gamma is not None else None
        be = beta.detach() if beta is not None else None
        if training:
            G = groups
            if F_ % G:
                raise RuntimeError(f"batch_norm_pool: {F_} frames do not split into {G} groups")
            rpg = M // G
            global _bn_stats_epoch
            _bn_stats_epoch += 1
            sums = _zero_scratch(owner, "_vc_sums", (2, G, C), dev)
            stat = torch.empty(4, G, C, device=dev, dtype=torch.float32)
            if partial is not None:
                L.call("vc_bn_stats_from_partial", L.ptr(partial), L.ptr(sums[0]), L.ptr(sums[1]), G, rpg // 64, C,
                       L.stream())
            else:
                L.call("vc_bn_stats", code, L.ptr(y), L.ptr(sums[0]), L.ptr(sums[1]), G, rpg, C, L.stream())
            L.call("vc_bn_finalize_count", L.ptr(sums[0]), L.ptr(sums[1]), L.ptr(ga), L.ptr(be), L.ptr(running_mean),
                   L.ptr(running_var), L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(stat[2]), L.ptr(stat[3]), G, C, rpg,
                   float(eps), float(momentum), int(running_mean is not None) | 2, L.ptr(_nbt(owner)), L.stream())
        else:
            G = 1
            stat = torch.empty(4, 1, C, device=dev, dtype=torch.float32)
            L.call("vc_bn_eval_params", L.ptr(ga), L.ptr(be), L.ptr(running_mean), L.ptr(running_var), L.ptr(stat[2]),
                   L.ptr(stat[3]), C, float(eps), L.stream())
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        out = torch.empty(N, D, Ho, Wo, C, device=dev, dtype=y.dtype)
        idx = torch.empty(N, D, Ho, Wo, C, device=dev, dtype=torch.uint8)
        L.call("vc_bn_pool_fwd", code, L.ptr(y), L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(out), L.ptr(idx), F_, H, W, C,
               k, s, p, F_ // G, L.stream())
        ctx.geom = (F_, H, W, C, k, s, p, G)
        ctx.training, ctx.owner, ctx.gamma, ctx.beta = training, owner, gamma, beta
        ctx.save_for_backward(y, stat, idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise RuntimeError("batch_norm_pool: backward through eval-mode BatchNorm is not implemented")
        y, stat, idx = ctx.saved_tensors
        F_, H, W, C, k, s, p, G = ctx.geom
        code = L.dtype_code(y.dtype)
        dout = cast(dout.contiguous(), y.dtype)
        sb = _zero_scratch(ctx.owner, "_vc_bsums", (2, G, C), y.device)  # cleared again by vc_bn_bwd_params
        dy = torch.empty_like(y)
        for phase in (0, 1):
            L.call("vc_bn_pool_bwd", code, phase, L.ptr(y), L.ptr(dout), L.ptr(idx), L.ptr(stat[0]), L.ptr(stat[1]),
                   L.ptr(stat[2]), L.ptr(stat[3]), L.ptr(sb[0]), L.ptr(sb[1]), L.ptr(dy), F_, H, W, C, k, s, p, G,
                   L.stream())
        dg = _grad_buf(ctx.gamma) if ctx.gamma is not None and ctx.gamma.requires_grad else None
        db = _grad_buf(ctx.beta) if ctx.beta is not None and ctx.beta.requires_grad else None
        L.call("vc_bn_bwd_params", L.ptr(sb[0]), L.ptr(sb[1]), L.ptr(dg), L.ptr(db), G, C, 3, L.stream())
        _ready(ctx.gamma, ctx.beta)
        return (dy,) + (None,) * 13


# VC_FUSE_BN_POOL=0 keeps BatchNorm + ReLU and the max-pool that follows as two passes each way (A/B measurements)
FUSE_BN_POOL = int(os.environ.get("VC_FUSE_BN_POOL", "1"))


def batch_norm_relu_pool(y, bn, groups, k, s, p=0, partial=None):
    """max_pool2d(relu(bn(y)), k, s, p) applied to every frame of y [N,D,H,W,C] (the max-pool never crosses D)."""
    training = bn.training or bn.running_mean is None
    # Overlapping windows (the ResNet stem's 3 / 2 / 1) re-read every input 2.25 times forward and gather up to four
    # windows per element backward: measured no faster than the two-pass form there, so only k == s is fused by
    # default (VC_FUSE_BN_POOL=2 fuses every geometry).
    if FUSE_BN_POOL == 0 or (k != s and FUSE_BN_POOL != 2):
        a = batch_norm(y, bn, groups, relu=True, partial=partial)
        N, D, H, W, C = a.shape
        o = max_pool2d(a.view(N * D, H, W, C), k, s, p)
        return o.view(N, D, o.shape[1], o.shape[2], C)
    out = _BatchNormPool.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, groups,
                               bn.momentum if bn.momentum is not None else 0.1, bn.eps, training, bn, partial, k, s, p)
    if training and bn.num_batches_tracked is not None and _nbt(bn) is None:
        bn.num_batches_tracked += groups
    return out


# ----------------------------------------------------------------------------------------------
# pooling
# ----------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        x = x.contiguous()
        N, H, W, C = x.shape
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = torch.empty(N, Ho, Wo, C, device=x.device, dtype=x.dtype)
        idx = torch.empty(N, Ho, Wo, C, device=x.device, dtype=torch.uint8)
        L.call("vc_maxpool2d_fwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(y), L.ptr(idx), N, H, W, C, k, s, p,
               L.stream())
        ctx.geom = (N, H, W, C, k, s, p)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, C, k, s, p = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(N, H, W, C, device=dy.device, dtype=dy.dtype)
        L.call("vc_maxpool2d_bwd", L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(idx), L.ptr(dx), N, H, W, C, k, s, p,
               L.stream())
        return dx, None, None, None


def max_pool2d(x, k, s, p=0):
    """x: [N,H,W,C] channels-last."""
    return _MaxPool.apply(x, k, s, p)


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        N, S, C = x.shape
        y = torch.empty(N, C, device=x.device, dtype=x.dtype)
        L.call("vc_avgpool_fwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(y), N, S, C, L.stream())
        ctx.geom = (N, S, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, S, C = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(N, S, C, device=dy.device, dtype=dy.dtype)
        L.call("vc_avgpool_bwd", L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(dx), N, S, C, L.stream())
        return dx


def global_avg_pool(x):
    """x: [N,S,C] -> [N,C] (mean over S)."""
    return _AvgPool.apply(x)


class _Relu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = elementwise(2, x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return elementwise(1, cast(dy.contiguous(), y.dtype), y)


def relu(x):
    return _Relu.apply(x)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return elementwise(0, a, b)

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return _Add.apply(a, b)


# ----------------------------------------------------------------------------------------------
# LSTM (multi-layer, optionally bidirectional), time-major
# ----------------------------------------------------------------------------------------------
def _lin_desc(M, K, N):
    return L.int_array([M, 1, 1, 1, K, 1, 1, 1, N, 1, 1, 1, 1, 1, 1, 0, 0, 0])


def _off(t, elems):
    """pointer to element offset `elems` inside tensor t"""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + elems * t.element_size())


# VC_PERSISTENT_LSTM=0 runs the recurrence as one small GEMM + one cell kernel per time step (A/B, and the
# path shapes the persistent kernel does not cover fall back to)
PERSISTENT_LSTM = os.environ.get("VC_PERSISTENT_LSTM", "1") != "0"
# VC_PERSISTENT_LSTM_BWD=0 keeps only the backward recurrence on the per-step kernels
PERSISTENT_LSTM_BWD = os.environ.get("VC_PERSISTENT_LSTM_BWD", "1") != "0"
_lstm_err = {}
# True while kernels of another stream may share the GPU with the step (RCCL bucket all-reduces overlapped with
# backward, VC_WGRAD_STREAM=1).  The persistent LSTM kernels need every one of their workgroups resident at once;
# beside foreign workgroups that is only taken for granted while the launch fills at most half of the CUs, larger
# launches go to the per-step kernels.
SIDE_STREAM_WORK = False


def lstm_persistent_allowed(query_name, code, B, H, dirs, device):
    """Whether a layer's recurrence may run as ONE persistent launch: the shape is supported, and if other streams
    may be active the launch leaves at least half of the CUs free."""
    if not PERSISTENT_LSTM or L.query(query_name, code, B, H, dirs) != 1:
        return False
    if SIDE_STREAM_WORK or WGRAD_STREAM:
        wgs = (H // 16) * dirs * ((B + 31) // 32)
        return 2 * wgs <= torch.cuda.get_device_properties(device).multi_processor_count
    return True


def lstm_error_flag(device):
    """int32 device word the persistent LSTM kernel sets if a workgroup gave up waiting for its peers."""
    key = (device.type, device.index)
    if key not in _lstm_err:
        _lstm_err[key] = torch.zeros(1, device=device, dtype=torch.int32)
    return _lstm_err[key]


def check_lstm_error():
    """Synchronising check of the persistent LSTM's error word (the trainer calls it where it reads loss and
    accuracy back anyway: every logging interval, the end of an epoch, after predictions).  The word is cleared
    before raising, so one time-out is reported once."""
    for flag in _lstm_err.values():
        if int(flag.item()) != 0:
            flag.zero_()
            raise RuntimeError("persistent LSTM kernel: a workgroup timed out waiting for its peers; the results "
                               "since the last check are invalid (VC_PERSISTENT_LSTM=0 selects the per-step kernels)")


class _LSTM(torch.autograd.Function):
    """x [T,B,In] -> out [T,B,H*dirs]; zero initial state (DecoderRNN passes None; TCG_LSTM uses the default).

    Per layer and direction: one MFMA GEMM hoists x W_ih^T + b for all T, then per step one small GEMM
    h_{t-1} W_hh^T and the fused cell kernel.  Backward is explicit BPTT with the same kernels; the weight
    gradients of all steps are one GEMM over the whole sequence."""

    @staticmethod
    def forward(ctx, x, H, layers, bidir, drop_p, *params):
        _require_cuda(x, "lstm")
        x = x.contiguous()
        T, B, In = x.shape
        dt = x.dtype
        code = L.dtype_code(dt)
        dirs = 2 if bidir else 1
        dev = x.device
        f32 = torch.float32
        saved = []
        masks = {}
        inp = x
        if In % 8:  # the bf16 GEMMs read 16-byte channel groups: zero columns up to a multiple of 8 (W_ih is packed alike)
            inp = torch.zeros(T, B, pad8(In), device=dev, dtype=dt)
            inp[:, :, :In] = x
        st = L.stream()
        persist = lstm_persistent_allowed("vc_lstm_persist_supported", code, B, H, dirs, dev)
        for l in range(layers):
            Kin = inp.shape[-1]
            out = torch.empty(T, B, H * dirs, device=dev, dtype=dt)
            pending = []
            for d in range(dirs):
                w_ih, w_hh, b_ih, b_hh = params[(l * dirs + d) * 4:(l * dirs + d) * 4 + 4]
                wih_f, wih_t = _packed_weights(w_ih, w_ih.shape[1], 1, Kin, 4 * H, dt)
                whh_f, whh_t = _packed_weights(w_hh, H, 1, H, 4 * H, dt)
                bsum = elementwise(0, b_ih.detach(), b_hh.detach())
                gx = torch.empty(T, B, 4 * H, device=dev, dtype=f32)
                L.call("vc_conv_fwd", _lin_desc(T * B, Kin, 4 * H), code, L.ptr(inp), 0, L.ptr(wih_f), L.ptr(bsum),
                       L.ptr(gx), 0, 0, 1, 0, st)
                c = torch.empty(T, B, H, device=dev, dtype=f32)
                act = torch.empty(T, B, 4 * H, device=dev, dtype=f32)
                if persist:  # the recurrence of every direction runs below as ONE persistent launch
                    pending.append((gx, whh_f, c, act))
                    saved.append((inp, c, act, wih_t, whh_t))
                    continue
                gh = torch.empty(B, 4 * H, device=dev, dtype=f32)
                order = range(T) if d == 0 else range(T - 1, -1, -1)
                prev = None
                for t in order:
                    if prev is not None:
                        L.call("vc_conv_fwd", _lin_desc(B, H, 4 * H), code, _off(out, (prev * B) * H * dirs + d * H),
                               H * dirs, L.ptr(whh_f), L.ptr(None), L.ptr(gh), 0, 0, 1, 0, st)
                    L.call("vc_lstm_cell_fwd", code, _off(gx, t * B * 4 * H), 4 * H, L.ptr(gh if prev is not None else None),
                           _off(c, prev * B * H) if prev is not None else L.ptr(None),
                           _off(out, (t * B) * H * dirs + d * H), H * dirs, _off(c, t * B * H),
                           _off(act, t * B * 4 * H), B, H, st)
                    prev = t
                saved.append((inp, c, act, wih_t, whh_t))
            if pending:
                import ctypes
                nb = L.query("vc_lstm_persist_xchg_bytes", B, H)
                xch = [torch.empty(nb, device=dev, dtype=torch.uint8) for _ in pending]
                arr = lambda ptrs: (ctypes.c_void_p * len(ptrs))(*ptrs)  # noqa: E731
                L.call("vc_lstm_persist_fwd", code, arr([p_[0].data_ptr() for p_ in pending]),
                       arr([p_[1].data_ptr() for p_ in pending]),
                       arr([out.data_ptr() + d * H * out.element_size() for d in range(dirs)]), H * dirs,
                       arr([p_[3].data_ptr() for p_ in pending]), arr([p_[2].data_ptr() for p_ in pending]),
                       arr([x_.data_ptr() for x_ in xch]), L.ptr(lstm_error_flag(dev)), T, B, H, dirs, st)
            saved.append(out)
            inp = out
            if drop_p > 0.0 and l < layers - 1:  # nn.LSTM's inter-layer dropout (training mode only)
                import ctypes
                inp = torch.empty_like(out)
                mask = torch.empty(out.shape, device=dev, dtype=torch.uint8)
                _dropout_seed[0] = (_dropout_seed[0] * 6364136223846793005 + 1442695040888963407) & 0x7FFFFFFFFFFFFFFF
                L.call("vc_dropout_fwd", code, L.ptr(out), L.ptr(inp), L.ptr(mask), L.i64(out.numel()), float(drop_p),
                       ctypes.c_uint64(_dropout_seed[0]), L.ptr(seed_counter(dev)), st)
                masks[l] = mask
        ctx.masks, ctx.drop_p = masks, drop_p
        ctx.meta = (T, B, In, H, layers, dirs)
        ctx.params = params
        ctx.saved_objs = saved
        return inp

    @staticmethod
    def backward(ctx, dout):
        T, B, In, H, layers, dirs = ctx.meta
        params = ctx.params
        saved = ctx.saved_objs
        dev = dout.device
        f32 = torch.float32
        st = L.stream()
        # unpack per-layer records: for each layer, `dirs` tuples then the layer output
        recs = []
        i = 0
        for l in range(layers):
            per_dir = [saved[i + d] for d in range(dirs)]
            out = saved[i + dirs]
            recs.append((per_dir, out))
            i += dirs + 1
        dt = recs[0][1].dtype
        code = L.dtype_code(dt)
        dlayer = cast(dout.contiguous(), dt)
        for l in range(layers - 1, -1, -1):
            per_dir, out = recs[l]
            inp = per_dir[0][0]
            Kin = inp.shape[-1]
            need_dinp = l > 0 or ctx.needs_input_grad[0]
            dinp = torch.empty(T, B, Kin, device=dev, dtype=dt) if need_dinp else None
            dgs = [torch.empty(T, B, 4 * H, device=dev, dtype=dt) for _ in range(dirs)]
            persist = PERSISTENT_LSTM_BWD and lstm_persistent_allowed("vc_lstm_persist_bwd_supported", code, B, H,
                                                                      dirs, dev)
            if persist:  # the recurrence of every direction of the layer as ONE persistent launch
                import ctypes
                nb = L.query("vc_lstm_persist_bwd_xchg_bytes", B, H)
                xch = [torch.empty(nb, device=dev, dtype=torch.uint8) for _ in range(dirs)]
                arr = lambda ptrs: (ctypes.c_void_p * len(ptrs))(*ptrs)  # noqa: E731
                L.call("vc_lstm_persist_bwd", code,
                       arr([dlayer.data_ptr() + d * H * dlayer.element_size() for d in range(dirs)]), H * dirs,
                       arr([per_dir[d][2].data_ptr() for d in range(dirs)]),
                       arr([per_dir[d][1].data_ptr() for d in range(dirs)]),
                       arr([per_dir[d][4].data_ptr() for d in range(dirs)]),
                       arr([g_.data_ptr() for g_ in dgs]), arr([x_.data_ptr() for x_ in xch]),
                       L.ptr(lstm_error_flag(dev)), T, B, H, dirs, st)
            for d in range(dirs):
                _, c, act, wih_t, whh_t = per_dir[d]
                w_ih, w_hh, b_ih, b_hh = params[(l * dirs + d) * 4:(l * dirs + d) * 4 + 4]
                dg = dgs[d]
                dh_rec = torch.empty(B, H, device=dev, dtype=f32)
                dc = [torch.empty(B, H, device=dev, dtype=f32), torch.empty(B, H, device=dev, dtype=f32)]
                order = list(range(T)) if d == 0 else list(range(T - 1, -1, -1))
                nxt = None  # the step processed after t in forward order (whose gradients flow back into t)
                for k in (range(T - 1, -1, -1) if not persist else ()):
                    t = order[k]
                    prev = order[k - 1] if k > 0 else None
                    if nxt is not None:
                        L.call("vc_conv_dgrad", _lin_desc(B, H, 4 * H), code, _off(dg, nxt * B * 4 * H), 0,
                               L.ptr(whh_t), L.ptr(dh_rec), 0, 1, 0, st)
                    L.call("vc_lstm_cell_bwd", code, _off(dlayer, (t * B) * H * dirs + d * H), H * dirs,
                           L.ptr(dh_rec if nxt is not None else None), L.ptr(dc[k & 1] if nxt is not None else None),
                           _off(act, t * B * 4 * H), _off(c, t * B * H),
                           _off(c, prev * B * H) if prev is not None else L.ptr(None), _off(dg, t * B * 4 * H),
                           L.ptr(dc[(k & 1) ^ 1]), B, H, st)
                    nxt = t
                # weight gradients over the whole sequence
                if w_ih.requires_grad and Kin == w_ih.shape[1]:
                    # a Linear weight's packed gradient layout is the parameter's own: accumulate in place
                    L.call("vc_conv_wgrad", _lin_desc(T * B, Kin, 4 * H), code, L.ptr(inp), 0, L.ptr(dg), 0,
                           L.ptr(_grad_buf(w_ih)), L.ptr(None), 0, st)
                elif w_ih.requires_grad:  # zero-padded input columns: packed rows are longer than the parameter's
                    dwp = _wgrad_scratch(w_ih, (4 * H, 1, Kin))
                    L.call("vc_conv_wgrad", _lin_desc(T * B, Kin, 4 * H), code, L.ptr(inp), 0, L.ptr(dg), 0,
                           L.ptr(dwp), L.ptr(None), 0, st)
                    L.call("vc_unpack_wgrad", L.ptr(dwp), L.ptr(_grad_buf(w_ih)), 4 * H, w_ih.shape[1], 1, Kin, 3, st)
                if w_hh.requires_grad and T > 1:
                    # pairs (h_prev, dgates_t): forward dir h[0..T-2] with dg[1..T-1]; reverse dir h[1..T-1] with dg[0..T-2]
                    h_off = 0 if d == 0 else 1
                    g_off = 1 if d == 0 else 0
                    L.call("vc_conv_wgrad", _lin_desc((T - 1) * B, H, 4 * H), code,
                           _off(out, (h_off * B) * H * dirs + d * H), H * dirs, _off(dg, g_off * B * 4 * H), 0,
                           L.ptr(_grad_buf(w_hh)), L.ptr(None), 0, st)
                for b_ in (b_ih, b_hh):
                    if b_.requires_grad:
                        L.call("vc_colsum", code, L.ptr(dg), L.ptr(_grad_buf(b_)), T * B, 4 * H, 4 * H, st)
                _ready(w_ih, w_hh, b_ih, b_hh)
                if need_dinp:
                    # the transposed pack has one row per TRUE input column: the padded columns of dinp (rows are Kin
                    # wide) are left unwritten and sliced off below
                    L.call("vc_conv_dgrad", _lin_desc(T * B, w_ih.shape[1], 4 * H), code, L.ptr(dg), 0, L.ptr(wih_t),
                           L.ptr(dinp), Kin, int(dt == torch.float32), int(d > 0), st)
            dlayer = dinp
            if l == 0 and dinp is not None and Kin != In:
                dlayer = dinp[:, :, :In].contiguous()
            if l > 0 and (l - 1) in ctx.masks:
                dlayer = torch.empty_like(dinp)
                L.call("vc_dropout_bwd", code, L.ptr(dinp), L.ptr(ctx.masks[l - 1]), L.ptr(dlayer), L.i64(dinp.numel()),
                       float(ctx.drop_p), st)
        return (dlayer, None, None, None, None) + (None,) * len(params)


def lstm(x_tm, lstm_module):
    """x_tm: [T,B,In] in the compute dtype; lstm_module: nn.LSTM holding the parameters (its inter-layer
    dropout is applied in training mode, with the library's own generator)."""
    m = lstm_module
    drop_p = float(m.dropout) if (m.training and m.num_layers > 1) else 0.0
    dirs = 2 if m.bidirectional else 1
    params = []
    for l in range(m.num_layers):
        for d in range(dirs):
            sfx = f"_l{l}" + ("_reverse" if d else "")
            params += [getattr(m, "weight_ih" + sfx), getattr(m, "weight_hh" + sfx), getattr(m, "bias_ih" + sfx),
                       getattr(m, "bias_hh" + sfx)]
    return _LSTM.apply(x_tm, m.hidden_size, m.num_layers, m.bidirectional, drop_p, *params)


# ----------------------------------------------------------------------------------------------
# cross-entropy head
# ----------------------------------------------------------------------------------------------
class _CrossEntropy(torch.autograd.Function):
    """mean cross-entropy over the batch (nn.CrossEntropyLoss default), fused with its gradient, the
    arg-max prediction and the number of correct predictions."""

    @staticmethod
    def forward(ctx, logits, target, num_classes, stats):
        _require_cuda(logits, "cross_entropy")
        if logits.dtype != torch.float32:
            raise RuntimeError("cross_entropy: logits must be float32")
        B = logits.shape[0]
        ld = logits.stride(0)
        tgt = target.reshape(-1).contiguous()
        loss = torch.zeros(1, device=logits.device, dtype=torch.float32)
        pred = torch.empty(B, device=logits.device, dtype=torch.int64)
        correct = torch.zeros(1, device=logits.device, dtype=torch.int32)
        dl = torch.empty(B, ld, device=logits.device, dtype=torch.float32)
        L.call("vc_ce_fwd_bwd", L.VC_F32, L.ptr(logits), ld, L.ptr(tgt), L.ptr(loss), L.ptr(pred), L.ptr(correct),
               L.ptr(dl), ld, B, num_classes, 1.0 / B, 1.0 / B, L.stream())
        ctx.save_for_backward(dl)
        ctx.C = logits.shape[1]
        if stats is not None:
            stats["pred"], stats["correct"] = pred, correct
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        # g is the 0-dim upstream gradient (1 for a plain loss.backward()); scaling [B, C] by it is plumbing
        d = dl[:, :ctx.C]
        return d * g, None, None, None


def cross_entropy(logits, target, stats=None):
    """logits [B,C] float32 (a view of a row-padded buffer is fine), target int64 [B] or [B,1]."""
    return _CrossEntropy.apply(logits, target, logits.shape[1], stats)


# ----------------------------------------------------------------------------------------------
# dropout
# ----------------------------------------------------------------------------------------------
_dropout_seed = [0x5EED]


def manual_seed(seed):
    """Seed of the dropout generator (counter-based; every call advances it)."""
    _dropout_seed[0] = int(seed) & 0x7FFFFFFFFFFFFFFF
    for c in _seed_counters.values():
        c.zero_()


# One int64 counter per device, folded into every dropout / drop-path seed at kernel execution time.  The
# fused optimizer advances it once per step, so a captured graph of the training step draws new masks on
# every replay although the host-side seed arguments are frozen in the graph.
_seed_counters = {}


def seed_counter(device):
    c = _seed_counters.get(device)
    if c is None:
        c = _seed_counters[device] = torch.zeros(1, device=device, dtype=torch.int64)
    return c


def advance_seed_counter(device):
    L.call("vc_counter_add", L.ptr(seed_counter(device)), L.i64(1), L.stream())


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        import ctypes
        x = x.contiguous()
        out = torch.empty_like(x)
        mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        _dropout_seed[0] = (_dropout_seed[0] * 6364136223846793005 + 1442695040888963407) & 0x7FFFFFFFFFFFFFFF
        L.call("vc_dropout_fwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(out), L.ptr(mask), L.i64(x.numel()), float(p),
               ctypes.c_uint64(_dropout_seed[0]), L.ptr(seed_counter(x.device)), L.stream())
        ctx.p = p
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.call("vc_dropout_bwd", L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(mask), L.ptr(dx), L.i64(dy.numel()),
               float(ctx.p), L.stream())
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, p)


# ----------------------------------------------------------------------------------------------
# TCG-LSTM fusion
# ----------------------------------------------------------------------------------------------
class _TCGInputs(torch.autograd.Function):
    """g [B,T,E], c [B,T,K,E] (fp32) -> short_in [T,B,2E] = [g | mean_K c], long_in [Tl,B,E] = g[:, ::down]."""

    @staticmethod
    def forward(ctx, g, c, down, dtype):
        _require_cuda(g, "tcg_inputs")
        g = g.contiguous().float()
        c = c.contiguous().float()
        B, T, E = g.shape
        K = c.shape[2]
        Tl = (T + down - 1) // down
        s_in = torch.empty(T, B, 2 * E, device=g.device, dtype=dtype)
        l_in = torch.empty(Tl, B, E, device=g.device, dtype=dtype)
        L.call("vc_tcg_inputs", L.dtype_code(dtype), L.ptr(g), L.ptr(c), L.ptr(s_in), L.ptr(l_in), B, T, Tl, K, E, down,
               L.stream())
        ctx.meta = (B, T, Tl, K, E, down)
        return s_in, l_in

    @staticmethod
    def backward(ctx, ds, dl):
        B, T, Tl, K, E, down = ctx.meta
        ds, dl = ds.contiguous(), dl.contiguous()
        dg = torch.empty(B, T, E, device=ds.device, dtype=torch.float32)
        dc = torch.empty(B, T, K, E, device=ds.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        L.call("vc_tcg_inputs_bwd", L.dtype_code(ds.dtype), L.ptr(ds), L.ptr(dl), L.ptr(dg), L.ptr(dc), B, T, Tl, K, E,
               down, L.stream())
        return dg, dc, None, None


def tcg_inputs(g, c, down, dtype):
    if c.dim() == 4 and c.stride(2) == 0:
        c = c[:, :, :1]  # k identical copies (an expanded view): their mean is any one of them
    return _TCGInputs.apply(g, c, down, dtype)


class _DropPath(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, samples):
        import ctypes
        x = x.contiguous()
        _dropout_seed[0] = (_dropout_seed[0] * 6364136223846793005 + 1442695040888963407) & 0x7FFFFFFFFFFFFFFF
        ctx.args = (float(p), ctypes.c_uint64(_dropout_seed[0]), x.numel() // samples)
        out = torch.empty_like(x)
        L.call("vc_drop_path", L.dtype_code(x.dtype), L.ptr(x), L.ptr(out), L.i64(x.numel()), L.i64(ctx.args[2]),
               ctx.args[0], ctx.args[1], L.ptr(seed_counter(x.device)), L.stream())
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.call("vc_drop_path", L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(dx), L.i64(dy.numel()), L.i64(ctx.args[2]),
               ctx.args[0], ctx.args[1], L.ptr(seed_counter(dy.device)), L.stream())
        return dx, None, None


def drop_path(x, p, training, samples):
    """Stochastic depth on a residual branch: x holds `samples` equal-sized contiguous samples."""
    if not training or p <= 0.0:
        return x
    return _DropPath.apply(x, p, samples)


class _TCGFuse(torch.autograd.Function):
    """h_s [T,B,2H], h_l [Tl,B,H] -> clip feature [B,2H] (interpolate, sigmoid gate, blend, softmax-over-time pool).
    The gate's Linear(4H -> 2H) runs on the MFMA GEMM; everything else is three fused kernels."""

    @staticmethod
    def forward(ctx, hs, hl, gate_w, gate_b):
        hs, hl = hs.contiguous(), hl.contiguous()
        T, B, F2 = hs.shape
        H = F2 // 2
        Tl = hl.shape[0]
        dt = hs.dtype
        code = L.dtype_code(dt)
        dev = hs.device
        st = L.stream()
        cat = torch.empty(T, B, 4 * H, device=dev, dtype=dt)
        L.call("vc_tcg_prep", code, L.ptr(hs), L.ptr(hl), L.ptr(cat), T, Tl, B, H, st)
        wf, wt = _packed_weights(gate_w, 4 * H, 1, 4 * H, 2 * H, dt)
        z = torch.empty(T, B, 2 * H, device=dev, dtype=torch.float32)
        L.call("vc_conv_fwd", _lin_desc(T * B, 4 * H, 2 * H), code, L.ptr(cat), 0, L.ptr(wf), L.ptr(gate_b.detach()),
               L.ptr(z), 0, 0, 1, 0, st)
        gate = torch.empty(T, B, 2 * H, device=dev, dtype=torch.float32)
        h = torch.empty(T, B, 2 * H, device=dev, dtype=torch.float32)
        alpha = torch.empty(B, T, device=dev, dtype=torch.float32)
        clip = torch.empty(B, 2 * H, device=dev, dtype=dt)
        L.call("vc_tcg_fuse_fwd", code, L.ptr(z), L.ptr(cat), L.ptr(gate), L.ptr(h), L.ptr(alpha), L.ptr(clip), T, B, H,
               st)
        ctx.meta = (T, Tl, B, H)
        ctx.gate_w, ctx.gate_b = gate_w, gate_b
        ctx.save_for_backward(cat, gate, h, alpha, wt)
        return clip

    @staticmethod
    def backward(ctx, dclip):
        cat, gate, h, alpha, wt = ctx.saved_tensors
        T, Tl, B, H = ctx.meta
        dt = cat.dtype
        code = L.dtype_code(dt)
        dev = cat.device
        st = L.stream()
        dclip = cast(dclip.contiguous(), dt)
        dz = torch.empty(T, B, 2 * H, device=dev, dtype=dt)
        dhs = torch.empty(T, B, 2 * H, device=dev, dtype=torch.float32)
        dup2 = torch.empty(T, B, 2 * H, device=dev, dtype=torch.float32)
        L.call("vc_tcg_fuse_bwd", code, L.ptr(dclip), L.ptr(cat), L.ptr(gate), L.ptr(h), L.ptr(alpha), L.ptr(dz),
               L.ptr(dhs), L.ptr(dup2), T, B, H, st)
        desc = _lin_desc(T * B, 4 * H, 2 * H)
        gw, gb = ctx.gate_w, ctx.gate_b
        if gw.requires_grad:
            L.call("vc_conv_wgrad", desc, code, L.ptr(cat), 0, L.ptr(dz), 0, L.ptr(_grad_buf(gw)),
                   L.ptr(_grad_buf(gb) if gb.requires_grad else None), 0, st)
        elif gb.requires_grad:
            L.call("vc_colsum", code, L.ptr(dz), L.ptr(_grad_buf(gb)), T * B, 2 * H, 2 * H, st)
        _ready(gw, gb)
        dcat = torch.empty(T, B, 4 * H, device=dev, dtype=dt)
        L.call("vc_conv_dgrad", desc, code, L.ptr(dz), 0, L.ptr(wt), L.ptr(dcat), 0, int(dt == torch.float32), 0, st)
        dhs_out = torch.empty(T, B, 2 * H, device=dev, dtype=dt)
        dhl = torch.empty(Tl, B, H, device=dev, dtype=dt)
        L.call("vc_tcg_combine_bwd", code, L.ptr(dcat), L.ptr(dhs), L.ptr(dup2), L.ptr(dhs_out), L.ptr(dhl), T, Tl, B, H,
               st)
        return dhs_out, dhl, None, None


def tcg_fuse(hs, hl, gate_linear):
    return _TCGFuse.apply(hs, hl, gate_linear.weight, gate_linear.bias)


# ----------------------------------------------------------------------------------------------
# Swin-T pieces: LayerNorm, GELU, shifted-window attention, patch merging
# ----------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _require_cuda(x, "layer_norm")
        x = x.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        out = torch.empty_like(x)
        stat = torch.empty(2, M, device=x.device, dtype=torch.float32)
        L.call("vc_layernorm_fwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(weight.detach()), L.ptr(bias.detach()),
               L.ptr(out), L.ptr(stat[0]), L.ptr(stat[1]), M, C, float(eps), L.stream())
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, stat)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, stat = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        dout = cast(dout.contiguous(), x.dtype)
        dx = torch.empty_like(x)
        w, b = ctx.weight, ctx.bias
        L.call("vc_layernorm_bwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(dout), L.ptr(w.detach()), L.ptr(stat[0]),
               L.ptr(stat[1]), L.ptr(dx), L.ptr(_grad_buf(w)), L.ptr(_grad_buf(b)), M, C, L.stream())
        _ready(w, b)
        return dx, None, None, None


def layer_norm(x, ln):
    """ln: nn.LayerNorm holding weight / bias / eps; normalises the last dimension."""
    return _LayerNorm.apply(x, ln.weight, ln.bias, ln.eps)


class _LayerNormFork(torch.autograd.Function):
    """(x, LayerNorm(x)): x goes on into the residual stream, its normalised copy into the branch.  The gradient
    of x is the sum of the two; the LayerNorm backward kernel adds the residual-stream gradient as it writes, so
    autograd never runs an addition pass of its own."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _require_cuda(x, "layer_norm")
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        out = torch.empty_like(x)
        stat = torch.empty(2, M, device=x.device, dtype=torch.float32)
        L.call("vc_layernorm_fwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(weight.detach()), L.ptr(bias.detach()),
               L.ptr(out), L.ptr(stat[0]), L.ptr(stat[1]), M, C, float(eps), L.stream())
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, stat)
        return x.view_as(x), out

    @staticmethod
    def backward(ctx, dres, dout):
        x, stat = ctx.saved_tensors
        if dout is None:
            return dres, None, None, None
        C = x.shape[-1]
        M = x.numel() // C
        dout = cast(dout.contiguous(), x.dtype)
        if dres is not None:
            dres = cast(dres.contiguous(), x.dtype)
        dx = torch.empty_like(x)
        w, b = ctx.weight, ctx.bias
        L.call("vc_layernorm_bwd_add", L.dtype_code(x.dtype), L.ptr(x), L.ptr(dout), L.ptr(w.detach()),
               L.ptr(stat[0]), L.ptr(stat[1]), L.ptr(dres), L.ptr(dx), L.ptr(_grad_buf(w)), L.ptr(_grad_buf(b)), M, C,
               L.stream())
        _ready(w, b)
        return dx, None, None, None


def layer_norm_fork(x, ln):
    """Returns (x, LayerNorm(x)) for a tensor that continues in the residual stream; see _LayerNormFork."""
    return _LayerNormFork.apply(x, ln.weight, ln.bias, ln.eps)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return elementwise(3, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return elementwise(4, cast(dy.contiguous(), x.dtype), x)


def gelu(x):
    return _Gelu.apply(x)


class _WindowAttention(torch.autograd.Function):
    """qkv [B*H*W, 3C] (token order, no window partition / roll materialised) -> attention output [B*H*W, C]."""

    @staticmethod
    def forward(ctx, qkv, bias_table, B, H, W, heads, ws, shift):
        _require_cuda(qkv, "window_attention")
        qkv = qkv.contiguous()
        C = qkv.shape[-1] // 3
        scale = (C // heads) ** -0.5
        out = torch.empty(qkv.shape[0], C, device=qkv.device, dtype=qkv.dtype)
        L.call("vc_window_attn_fwd", L.dtype_code(qkv.dtype), L.ptr(qkv), L.ptr(bias_table.detach()), L.ptr(out), B, H, W,
               C, heads, ws, shift, float(scale), L.stream())
        ctx.meta = (B, H, W, C, heads, ws, shift, scale)
        ctx.table = bias_table
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        B, H, W, C, heads, ws, shift, scale = ctx.meta
        dout = cast(dout.contiguous(), qkv.dtype)
        dqkv = torch.empty_like(qkv)
        tb = ctx.table
        dtb = _grad_buf(tb) if tb.requires_grad else None
        L.call("vc_window_attn_bwd", L.dtype_code(qkv.dtype), L.ptr(qkv), L.ptr(dout), L.ptr(tb.detach()), L.ptr(dqkv),
               L.ptr(dtb), B, H, W, C, heads, ws, shift, float(scale), L.stream())
        _ready(tb)
        return dqkv, None, None, None, None, None, None, None


def window_attention(qkv, bias_table, B, H, W, heads, ws, shift):
    return _WindowAttention.apply(qkv, bias_table, B, H, W, heads, ws, shift)


class _PatchMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, H, W, C = x.shape
        out = torch.empty(B, H // 2, W // 2, 4 * C, device=x.device, dtype=x.dtype)
        L.call("vc_patch_merge", L.dtype_code(x.dtype), L.ptr(x), L.ptr(out), B, H, W, C, 0, L.stream())
        ctx.shape = (B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C, device=dy.device, dtype=dy.dtype)
        L.call("vc_patch_merge", L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(dx), B, H, W, C, 1, L.stream())
        return dx


def patch_merge(x):
    """[B,H,W,C] -> [B,H/2,W/2,4C] with timm's PatchMerging channel order."""
    return _PatchMerge.apply(x)
