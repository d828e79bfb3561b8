#!/usr/bin/env python3
"""Throughput benchmark of the hot path: train clips/sec on N MI355X of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): ResNetCRNN =
ResCNNEncoder (bottleneck ResNet-50 trunk, heads 1024/768/512) + DecoderRNN (2-layer LSTM, h=512, FC 256,
101 classes), batch 32 clips per GPU of 16 x 3 x 224 x 224, bf16 compute with fp32 master weights, one
full training step = forward + cross-entropy + backward + gradient all-reduce (N>1) + fused Adam.
`--arch resnet152 --lstm-layers 3` selects what the reference's code actually instantiates.
Clips are synthetic (randn, seed 1234+rank), weights are the modules' default initialisation
(seed 0) and are resident in HBM before the timed region.  One rank per GPU, weak scaling (32 clips
per GPU whatever N).

The JSON line also carries
  roofline     : the forward implicit-GEMM convolution kernel (the dominant kernel family) -- algorithmic
                 FLOP/s (2*M*Cout*Cin*taps per launch, unpadded channels, summed over the launches of the
                 timed region / their summed durations from HIP events on the launch stream) against the dense
                 bf16 MFMA peak of 2.5 PFLOP/s.
  cpu_baseline : the CPU oracle (oracle/torch_oracle.py, same architecture, fp32) timed on this host for a
                 bounded sample (5 steps of 4 clips, the configs[0] batch), rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X
HBM_PEAK_GBS = 8000.0           # HBM3E peak, MI355X


def build_workload(args, M, S, oracle=False):
    """(model, clip shape per GPU, workload description) for --model; `oracle` builds the CPU restatement."""
    O = None
    if oracle:
        from oracle import torch_oracle as O
    B = args.batch
    if args.model == "resnetcrnn":
        cfg = model_config(args)
        model = (O or M).ResNetCRNNModel(cfg)
        return model, (B, 16, 3, 224, 224), (
            f"ResNetCRNN training step: ResCNNEncoder({args.arch}, 1024/768/512) + DecoderRNN({args.lstm_layers}x512 "
            f"LSTM, FC 256, 101 classes), {B} clips/GPU x 16 x 3 x 224 x 224, dropout {args.dropout}")
    if args.model == "crnn":
        cfg = types.SimpleNamespace(img_x=224, img_y=224, CNN_fc_hidden1=1024, CNN_fc_hidden2=768, CNN_embed_dim=512,
                                    dropout_p=0.0, use_pretrained=False, RNN_hidden_layers=3, RNN_hidden_nodes=512,
                                    RNN_FC_dim=256, num_classes=101)
        return (O or M).CRNNModel(cfg), (B, 16, 3, 224, 224), (
            f"CRNN training step: EncoderCNN(1024/768/512) + DecoderRNN(3x512), {B} clips/GPU x 16 x 3 x 224 x 224")
    if args.model == "conv3d":
        cfg = types.SimpleNamespace(dropout_p=0.5, num_classes=101)
        return (O or M).Conv3DModel(cfg), (B, 1, 16, args.conv3d_size, args.conv3d_size), (
            f"Conv3D training step: Conv3DModel, {B} clips/GPU x 1 x 16 x {args.conv3d_size} x {args.conv3d_size}, dropout 0.5")
    if args.model == "swinrnn":
        enc = (O or S).SwinTransformerEncoder(1024, 768, 0.0, 512)
        dec = (O or M).DecoderRNN(512, args.lstm_layers, 512, 256, 0.0, 101)
        return (O or M).SwinTransformerRNNModel(enc, dec), (B, 16, 3, 224, 224), (
            f"swintransformer-RNN training step: Swin-T encoder (drop_path 0.1, 1024/768/512 head) + DecoderRNN({args.lstm_layers}x512), "
            f"{B} clips/GPU x 16 x 3 x 224 x 224")
    if args.model == "clusterswin":
        enc = (O or S).ClusterSwinEncoder(embed_dim=96, k_tokens=8)
        dec = (O or M).TCG_LSTM(96, 512, 2, 8, 4, 101, 0.1)
        return (O or M).SwinTCGModel(enc, dec), (B, 16, 3, 224, 224), (
            f"clusterSwin_TCGlstm training step: ClusterSwinEncoder(Swin-T, drop_path 0.1, E=96, k=8) + TCG_LSTM(H=512, L=2, drop 0.1), "
            f"{B} clips/GPU x 16 x 3 x 224 x 224")
    raise SystemExit(f"unknown --model {args.model}")


def model_config(args):
    return types.SimpleNamespace(CNN_fc_hidden1=1024, CNN_fc_hidden2=768, CNN_embed_dim=512, dropout_p=args.dropout,
                                 use_pretrained=False, RNN_hidden_layers=args.lstm_layers, RNN_hidden_nodes=512,
                                 RNN_FC_dim=256, num_classes=101, arch=args.arch)


def cpu_baseline(args):
    """The oracle's training step on the host CPU, bounded sample."""
    import torch.nn as nn
    from oracle import torch_oracle as O
    b = args.cpu_batch
    torch.manual_seed(0)
    model, shape, _ = build_workload(args, None, None, oracle=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    crit = nn.CrossEntropyLoss()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(b, *shape[1:], generator=g)
    y = torch.randint(0, 101, (b, 1), generator=g)
    O.train_step(model, x, y, crit, opt)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < args.cpu_steps:
        O.train_step(model, x, y, crit, opt)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": b * n / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} training steps of {b} clips {tuple(shape[1:])} after 1 warm-up, fp32, same architecture"}


def spawn_ranks(n):
    """One child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them),
    same command line; rank 0 prints the JSON line.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            code = p.wait()
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in procs:  # a rank died: the others would wait in a collective forever
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


def trace(msg):
    """Progress marks on stderr (VC_BENCH_TRACE=1): where a multi-rank run is, should it stall."""
    if os.environ.get("VC_BENCH_TRACE", "0") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="resnetcrnn", choices=["resnetcrnn", "crnn", "conv3d", "swinrnn", "clusterswin"])
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default 32; 8 for the Swin models)")
    ap.add_argument("--conv3d-size", type=int, default=112)
    ap.add_argument("--arch", default="resnet50", choices=["resnet50", "resnet101", "resnet152"])
    ap.add_argument("--lstm-layers", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.3, help="configs/ResNetCRNN_train.yaml model.dropout")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--cpu-batch", type=int, default=4, help="clips per CPU-baseline step (4 = the configs[0] batch)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--infer", action="store_true",
                    help="time the eval-mode forward (predict.py / save_predictions callers) instead of a training step")
    ap.add_argument("--deterministic", action="store_true",
                    help="trainer.set_deterministic(): ordered sums instead of fp32 atomics, steps repeat bit for bit")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 8 if args.model in ("swinrnn", "clusterswin") else 32

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly as `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the
        # GPU and never will; it starts one fresh rank process per GPU and ends with their worst exit code.
        raise SystemExit(spawn_ranks(args.gpus))

    from video_classification_amd import modules as M
    from video_classification_amd import swin as S
    from video_classification_amd import ops
    from video_classification_amd import trainer as TR

    rank, local_rank, world = TR.init_distributed()
    trace(f"distributed initialised: world {world}")
    if os.environ.get("VC_BENCH_TRACE", "0") == "1":
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("VC_BENCH_TRACE_AFTER", "90")), exit=False)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.deterministic:
        TR.set_deterministic(0)
    torch.manual_seed(0)
    model, shape, workload = build_workload(args, M, S)
    if args.deterministic:
        workload += ", deterministic mode"
    model = model.to(dev).set_compute_dtype(dt)
    opt = TR.get_optimizer(model, "adam", lr=1e-4, weight_decay=1e-4)
    trace("model and optimizer built")
    bucketer = TR.make_data_parallel(model, opt, bucket_mb=args.bucket_mb)
    trace("parameters broadcast")
    crit = TR.FusedCrossEntropyLoss()
    ops.manual_seed(1234 + rank)
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    x = torch.randn(*shape, generator=g).to(dev)
    y = torch.randint(0, 101, (B, 1), generator=g).to(dev)
    if args.infer:
        # forward only, eval-mode BatchNorm, replayed as one hipGraph; every rank runs its own replica
        from video_classification_amd.graph import GraphedInference
        infer = GraphedInference(model, x)
        for _ in range(args.warmup):
            infer(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            infer(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        if rank == 0:
            print(json.dumps({"metric": "inference clips/sec (16x224^2 UCF101 shape)", "value": world * B * args.steps / el,
                              "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                              "config": {"workload": workload.replace("training step", "eval-mode forward"),
                                         "global_batch": world * B, "launch": "hipgraph"}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    model.train()

    def eager_step():
        opt.zero_grad()
        out = model(x)
        loss = crit(out, y)
        loss.backward()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Step executor: the whole step captured once into a hipGraph and replayed (default), or launched kernel by
    # kernel from Python (--no-graph, and the fallback if capture is refused).  Same kernels, same work.
    mode = "eager"
    step = eager_step
    if not args.no_graph:
        try:
            from video_classification_amd.graph import GraphedTrainStep
            gstep = GraphedTrainStep(model, opt, crit, x, y, warmup=max(1, min(args.warmup, 3)))
            step = lambda: gstep(gstep.x, gstep.y)  # noqa: E731  (inputs already resident in the static buffers)
            mode = "hipgraph"
        except Exception as exc:  # pragma: no cover - depends on the runtime
            print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
    trace(f"step executor: {mode}")
    for _ in range(args.warmup):
        step()
    trace("warm-up done")

    if mode == "eager":
        ops.PROFILE = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.check_lstm_error()  # a persistent LSTM kernel that timed out would have made the timed steps meaningless
    trace("timed region done")
    prof_steps = args.steps
    if mode == "eager":
        prof, ops.PROFILE = ops.PROFILE, None
        prof_where = "HIP events around every launch inside the timed region"
    else:
        # events cannot be recorded inside a replayed graph: time the same kernels in eager steps right after
        prof_steps = 2
        ops.PROFILE = []
        for _ in range(prof_steps):
            eager_step()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        prof_where = f"HIP events around every launch in {prof_steps} eager steps run right after the timed region"
    # what every rank exchanged: [rank, world as the communicator sees it, all-reduces issued, buckets that had to be
    # launched by the optimizer's wait instead of from inside the backward pass (not overlapped)]
    mine = [rank, dist.get_world_size() if dist.is_initialized() else 1, bucketer.launches,
            getattr(bucketer, "total_late", 0)]
    exchange = [mine]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        rows = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(mine, dtype=torch.int64, device=dev))
        exchange = [r.tolist() for r in rows]

    dump = os.environ.get("VC_BENCH_DUMP_LAUNCHES")
    if dump and rank == 0:  # one line per contraction launch of the profiled steps, in launch order
        with open(dump, "w") as fh:
            for tag, flops, s, e, nbytes in prof:
                fh.write(f"{tag} {flops / 1e9:.2f} GF {nbytes / 1e6:.1f} MB {s.elapsed_time(e) * 1e3:.1f} us\n")
    # per-kernel-family time from the HIP events recorded around each contraction launch
    fam = {}
    for tag, flops, s, e, nbytes in prof:
        f = fam.setdefault(tag, [0.0, 0.0, 0, 0.0])
        f[0] += flops
        f[1] += s.elapsed_time(e) * 1e-3
        f[2] += 1
        f[3] += nbytes
    roof = None
    if "conv_fwd" in fam and fam["conv_fwd"][1] > 0:
        fl, tm, n, nb = fam["conv_fwd"]
        ach = fl / tm / 1e12
        # the family mixes MFMA-bound 3x3 layers with HBM-bound 1x1 / stem layers: the time each launch would take
        # at whichever of the two peaks binds it, summed, against the measured time of the family
        floor_s = sum(max(f_ / (MFMA_BF16_PEAK_TFLOPS * 1e12), b_ / (HBM_PEAK_GBS * 1e9))
                      for t_, f_, _, _, b_ in prof if t_ == "conv_fwd")
        # HBM bytes per launch of the same kernel family from the PMC passes (tools/pmc_bench.sh, separate runs of
        # this command with --no-graph; FETCH_SIZE doubled as gfx950 requires) -- None if they were never collected
        traffic, traffic_src, traffic_step, traffic_commit = None, None, None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))  # newest file last
        tfile = tfiles[-1] if tfiles else ""
        if args.model == "resnetcrnn" and args.arch == "resnet50" and B == 32 and os.path.exists(tfile):
            with open(tfile) as fh:
                tj = json.load(fh)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from hbm_traffic import kernel_sources_digest
            traffic_commit = tj.get("commit")
            if tj.get("kernel_sources_sha256") != kernel_sources_digest(ROOT):
                # counters taken on other kernel sources than the ones this run was built from: not this run's traffic
                traffic_src = (f"profiles/{os.path.basename(tfile)} was measured on other kernel sources (commit "
                               f"{traffic_commit}): re-run tools/pmc_bench.sh + tools/hbm_traffic.py")
            elif "conv_fwd" in tj.get("families", {}):
                traffic = tj["families"]["conv_fwd"]["traffic_bytes"]
                traffic_step = tj["families"]["conv_fwd"].get("traffic_bytes_per_step")
                traffic_src = f"profiles/{os.path.basename(tfile)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
        roof = {"bound": "mfma", "kernel": "igemm_nt_persist_kernel<0,*> (forward implicit-GEMM conv/linear)",
                "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_BF16_PEAK_TFLOPS, "frac_of_per_launch_roofline": floor_s / tm,
                "per_launch_roofline": "sum over launches of max(flops / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) / "
                                       "measured time of the family",
                "traffic": traffic, "traffic_unit": "bytes per launch",
                "traffic_source": traffic_src, "traffic_commit": traffic_commit, "traffic_bytes_per_step": traffic_step,
                "algorithmic_bytes_per_launch": nb / n, "algorithmic_bytes_per_step": nb / prof_steps,
                "launches": n, "avg_launch_ms": tm / n * 1e3, "avg_launch_gflop": fl / n / 1e9,
                "other_kernels": {k: {"achieved_tflops": v[0] / v[1] / 1e12, "launches": v[2],
                                      "ms_per_step": v[1] / prof_steps * 1e3}
                                  for k, v in fam.items() if v[1] > 0},
                "measured": prof_where}

    # whole-step figure of the metric's "% MFMA peak": algorithmic FLOPs of the model (3 x forward for a training
    # step) at the measured rate, per GPU, against the dense bf16 peak
    from video_classification_amd import flops as FL
    fwd_flops = FL.flops_per_clip(model, 16, shape[-2], shape[-1])
    step_tflops = 3.0 * fwd_flops * B * args.steps / elapsed / 1e12
    whole_step = {"forward_gflop_per_clip": fwd_flops / 1e9, "train_tflops_per_gpu": step_tflops,
                  "frac_of_mfma_peak": step_tflops / MFMA_BF16_PEAK_TFLOPS,
                  "note": "3 x forward FLOPs (convolution, linear, LSTM and attention products) x clips/s per GPU"}
    if rank == 0:
        line = {
            "metric": "train clips/sec (16x224^2 UCF101 shape)",
            "value": world * B * args.steps / elapsed,
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": workload + ", fwd+CE+bwd+all-reduce+Adam",
                       "global_batch": world * B, "frames_per_clip": 16, "parallelism": f"dp{world}",
                       "launch": mode},
            "exchange": {"backend": dist.get_backend() if dist.is_initialized() else None,
                         "buckets": len(bucketer.buckets), "bucket_mb": args.bucket_mb,
                         "per_rank": [{"rank": r, "comm_world": w, "allreduces_issued": n, "late_buckets": late}
                                      for r, w, n, late in exchange],
                         "note": "late_buckets = gradient buckets whose all-reduce was not started from inside the "
                                 "backward pass (0 = every exchange overlapped)"},
            "loss": float(loss.detach()),
            "roofline": roof,
            "whole_step": whole_step,
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
