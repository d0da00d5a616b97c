import torch, time
dev=torch.device("cuda")
for n in (2**28,):   # 1 GiB fp32
    x=torch.randn(n,device=dev); y=torch.empty_like(x)
    for fn,name,bytes_ in ((lambda: y.copy_(x),"copy",8*n),(lambda: y.zero_(),"fill",4*n),(lambda: torch.add(x,1.0,out=y),"add1",8*n),(lambda: x.sum(),"sum(read)",4*n)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print(f"{name}: {bytes_/ms/1e9:.2f} TB/s ({ms*1e3:.0f} us)")
