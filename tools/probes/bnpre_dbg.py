import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import golden_util as G
from video_classification_amd import modules as M, ops
cuda = torch.device("cuda:0")
torch.manual_seed(5)
T, B = 2, 16
trunk = M.make_resnet_trunk("resnet50", width=64)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
blocks = torch.nn.Sequential(*list(trunk[4])[:nb]).to(cuda)
x = torch.randn(T * B, 1, 16, 16, 64, device=cuda).bfloat16()
go = torch.randn(T * B, 1, 16, 16, 256, device=cuda).bfloat16()
grads = {}
for fused in (True, False, "again", "fused2"):
    ops.FUSE_BN_PRE = fused in (True, 'fused2')
    blocks.zero_grad(set_to_none=True)
    xi = x.clone().requires_grad_(True)
    a = xi
    for blk in blocks:
        a = blk.run(a, T)
    a.backward(go.clone())  # the last block owns the gradient it is handed and overwrites it
    torch.cuda.synchronize()
    grads[fused] = {n: p.grad.float().clone() for n, p in blocks.named_parameters()}
    grads[fused]["x"] = xi.grad.float().clone()
import statistics
for a, b in ((True, False), ("again", False), (True, "fused2"), ("fused2", "again")):
    e = [G.max_rel(grads[a][n].cpu(), grads[b][n].cpu()) for n in grads[True]]
    print(a, b, "max %.3e median %.3e" % (max(e), statistics.median(e)))
