import torch, time
x = torch.randn(4,1,384,128,128).pin_memory(); y = torch.randn(4,1,384,128,128).pin_memory()
xp = torch.randn(4,1,384,128,128)
torch.cuda.synchronize()
for name,(a,b) in {"pinned":(x,y),"pageable":(xp,xp)}.items():
    for _ in range(2): a.cuda(non_blocking=True); b.cuda(non_blocking=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): a.cuda(non_blocking=True); b.cuda(non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(name, f"{dt*1e3:.2f} ms per batch-4 image+label ({2*a.numel()*4/dt/1e9:.1f} GB/s)")
