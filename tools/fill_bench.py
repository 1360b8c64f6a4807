import torch, time
n=20000
a=torch.empty((n,n),dtype=torch.float64,device='cuda')
b=torch.empty((n,n),dtype=torch.float64,device='cuda')
for name,fn in (("fill",lambda: a.fill_(1.5)),("copy",lambda: b.copy_(a)),("exp_",lambda: b.exp_()),("mul",lambda: torch.mul(a,2.0,out=b))):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/5
    print(f"{name}: {ms*1e3:.0f} us  -> {8*n*n/ms/1e6:.0f} GB/s per 3.2GB pass")
