import torch, time
dev='cuda'
n=1<<29  # 2^29 bf16 = 1 GiB
a=torch.empty(n,device=dev,dtype=torch.bfloat16); b=torch.empty_like(a)
def t(f,reps=10):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
gb=n*2/1e9
print('fill   %.3f ms  %.2f TB/s (W)'%(t(lambda:a.zero_()), gb/t(lambda:a.zero_())))
print('copy   %.3f ms  %.2f TB/s (R+W)'%(t(lambda:b.copy_(a)), 2*gb/t(lambda:b.copy_(a))))
print('sum    %.3f ms  %.2f TB/s (R)'%(t(lambda:a.view(torch.int16).sum()), gb/t(lambda:a.view(torch.int16).sum())))
c=torch.empty(n//4,device=dev,dtype=torch.bfloat16)
print('expand 1R:4W  %.3f ms  %.2f TB/s'%(t(lambda:torch.cat([c,c,c,c],out=a)), (gb+gb)/t(lambda:torch.cat([c,c,c,c],out=a))))
