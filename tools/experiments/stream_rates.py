import torch
dev='cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
for mb in (16,64,150,600,2000):
    x=torch.randn(mb*1024*1024//4,device=dev); y=torch.empty_like(x)
    a=t(lambda: torch.sum(x)); b=t(lambda: y.copy_(x)); c=t(lambda: y.zero_())
    print("%5d MB: sum %.1f us (%.2f TB/s read)  copy %.1f us (%.2f TB/s r+w)  memset %.1f us (%.2f TB/s write)"%(mb,a,mb*1.048576/a,b,2*mb*1.048576/b,c,mb*1.048576/c))
