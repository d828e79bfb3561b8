import sys, torch, torch.nn as nn
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import golden_util as G
from oracle import torch_oracle as O
from video_classification_amd import modules as M
cuda = torch.device('cuda:0')
NB = int(sys.argv[1])
torch.manual_seed(31)
def stage(mod):
    ds = nn.Sequential(nn.Conv2d(256, 512, 1, stride=2, bias=False), nn.BatchNorm2d(512))
    return nn.Sequential(*([mod.Bottleneck(256, 128, 2, ds), mod.Bottleneck(512, 128), mod.Bottleneck(512, 128), mod.Bottleneck(512, 128)][:NB]))
ref = stage(O)
for m in ref.modules():
    if isinstance(m, nn.Conv2d): nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.uniform_(m.weight, 0.5, 1.5); nn.init.uniform_(m.bias, -0.3, 0.3)
for blk in ref: nn.init.uniform_(blk.bn3.weight, 0.2, 0.4)
B, T, S = 4, 2, 56
g = torch.Generator().manual_seed(32)
x = torch.randn(T * B, 256, S, S, generator=g).relu().bfloat16().float()
dy = torch.randn(T * B, 512, S // 2, S // 2, generator=g).bfloat16().float()
def run_ref(emu):
    ref.train(); ref.zero_grad()
    xr = x.clone().requires_grad_(True)
    O.EMULATE_BF16 = emu
    out_r = torch.cat([ref(xr[t * B:(t + 1) * B]) for t in range(T)]); out_r.backward(dy)
    O.EMULATE_BF16 = False
    d = {k: p.grad.clone() for k, p in ref.named_parameters()}; d['x'] = xr.grad.clone(); d['out']=out_r.detach()
    return d
def run_hip(dt):
    mine = stage(M); mine.load_state_dict(ref.state_dict(), strict=True); mine.to(cuda).train()
    xm = x.permute(0, 2, 3, 1).contiguous().view(T * B, 1, S, S, 256).to(cuda, dt).requires_grad_(True)
    a = xm
    for blk in mine: a = blk.run(a, T)
    dym = dy.permute(0, 2, 3, 1).contiguous().view(T * B, 1, S // 2, S // 2, 512).to(cuda, dt)
    a.backward(dym); torch.cuda.synchronize()
    d = {k: p.grad.detach().cpu() for k, p in mine.named_parameters()}
    d['x'] = xm.grad.detach().float().cpu().view(T * B, S, S, 256).permute(0, 3, 1, 2)
    d['out'] = a.detach().float().cpu().view(T * B, S // 2, S // 2, 512).permute(0, 3, 1, 2)
    return d
r32, remu = run_ref(False), run_ref(True)
h32, h16 = run_hip(torch.float32), run_hip(torch.bfloat16)
def cmp(a, b, name):
    errs = {k: G.l2_rel(a[k], b[k]) for k in a}
    w = max(errs, key=errs.get)
    print(f"{name:28s} worst {w} {errs[w]:.3e} median {sorted(errs.values())[len(errs)//2]:.3e} out {errs['out']:.3e} x {errs['x']:.3e} 0.conv3 {errs['0.conv3.weight']:.3e} 0.bn3.w {errs['0.bn3.weight']:.3e} 0.conv1 {errs['0.conv1.weight']:.3e}")
cmp(h32, r32, "hip f32 vs oracle f32")
cmp(remu, r32, "oracle emu vs oracle f32")
cmp(h16, r32, "hip bf16 vs oracle f32")
cmp(h16, remu, "hip bf16 vs oracle emu")
h16b = run_hip(torch.bfloat16)
cmp(h16b, h16, "hip bf16 run2 vs run1")
