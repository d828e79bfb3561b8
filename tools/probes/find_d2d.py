"""Which Python call sites issue the small device-to-device copies of an eager training step (torch.profiler, stacks)."""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from video_classification_amd import modules as M, ops, swin as S, trainer as TR

args = argparse.Namespace(model="resnetcrnn", batch=8, conv3d_size=112, arch="resnet50", lstm_layers=2, dropout=0.3)
dev = torch.device("cuda:0")
model, shape, _ = bench.build_workload(args, M, S)
model = model.to(dev).set_compute_dtype(torch.bfloat16)
model.train()
opt = None
x = torch.randn(8, *shape[1:], device=dev) if len(shape) > 1 else None
y = torch.randint(0, 101, (8, 1), device=dev)
def step():
    out = model(x)
    loss = ops.cross_entropy(out, y)
    loss.backward()
    if opt is not None:
        opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if "copy" in n.lower() or "Memcpy" in n or "fill" in n.lower() or "Memset" in n:
        st = [s for s in (e.stack or []) if "video-classification_amd" in s or "bench" in s or "trainer" in s]
        cnt[(n[:40], st[0] if st else "?")] += 1
for k, v in cnt.most_common(30):
    print(v, k)
