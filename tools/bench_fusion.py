"""Times ImageDensityFusion (forward + backward) on the device: python tools/bench_fusion.py [frames] [size]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_classification_amd import fusion  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = int(sys.argv[2]) if len(sys.argv) > 2 else 224
dev = torch.device("cuda:0")
m = fusion.create_image_density_fusion().to(dev)
m.train()
x = torch.randn(n, 3, s, s, device=dev)
d = torch.rand(n, 1, 32, 32, device=dev)


def step():
    m.zero_grad(set_to_none=False)
    m(x, d).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.time()
reps = 5
for _ in range(reps):
    step()
torch.cuda.synchronize()
ms = (time.time() - t) / reps * 1e3
fl = 2.0 * n * s * s * (4 * 64 * 9 + 2 * 64 * 64 * 9 + 64 * 32 * (1 + 9 + 25 + 49) + 128 * 128 + 128 * 512) * 3
print(f"ImageDensityFusion fwd+bwd, {n} frames of {s}x{s}: {ms:.2f} ms, {n / ms * 1e3:.0f} frames/s, "
      f"{fl / ms / 1e9:.0f} TFLOP/s on the convolutions")
