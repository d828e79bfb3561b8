"""Which stock ATen operators still launch kernels inside a training step: one eager step of the bench workload under
torch.profiler, operators with device time listed with their input shapes and Python call sites.
usage: python tools/probes/aten_in_step.py [model]"""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    model_name = sys.argv[1] if len(sys.argv) > 1 else "resnetcrnn"
    from video_classification_amd import modules as M
    from video_classification_amd import swin as S
    from video_classification_amd import ops
    from video_classification_amd import trainer as TR
    args = argparse.Namespace(model=model_name, batch=8 if model_name in ("swinrnn", "clusterswin") else 32,
                              conv3d_size=112, arch="resnet50", lstm_layers=2, dropout=0.3, dtype="bf16")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model, shape, _ = bench.build_workload(args, M, S)
    model = model.to(dev).set_compute_dtype(torch.bfloat16)
    opt = TR.get_optimizer(model, "adam", lr=1e-4, weight_decay=1e-4)
    crit = TR.FusedCrossEntropyLoss()
    ops.manual_seed(1)
    x = torch.randn(*shape).to(dev)
    y = torch.randint(0, 101, (args.batch, 1)).to(dev)
    model.train()

    def step():
        opt.zero_grad()
        loss = crit(model(x), y)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    # device-to-device copies and fills that ATen issues as memcpy / memset (no kernel of its own): by call site
    memops = {}
    for ev in prof.events():
        if ev.name.startswith("aten::copy_") or ev.name.startswith("aten::zero_") or ev.name.startswith("aten::fill_"):
            if any(k.name.lower().startswith(("memcpy", "memset")) for k in ev.kernels):
                site = [s for s in ev.stack if "video-classification_amd" in s or "video_classification_amd" in s]
                key = (ev.name, str(ev.input_shapes)[:40], (site[0] if site else "?")[-80:])
                c = memops.setdefault(key, [0, 0.0])
                c[0] += 1
                c[1] += sum(k.duration for k in ev.kernels)
    for key, c in sorted(memops.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"memop {key[0]:14s} x{c[0]:4d} {c[1]:8.1f} us {key[1]:40s} {key[2]}")
    rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
            if e.key.startswith("aten::") and getattr(e, "self_device_time_total", 0) > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:40]:
        stack = [s for s in e.stack if "video-classification_amd" in s or "video_classification_amd" in s or "bench" in s]
        print(f"{e.key:28s} x{e.count:4d} {e.self_device_time_total:9.1f} us  {str(e.input_shapes)[:70]:70s} "
              f"{(stack[0] if stack else '')[-90:]}")


if __name__ == "__main__":
    main()
