import sys; sys.path.insert(0,'.')
import torch
from gptst_amd import ops
dev='cuda:0'
def t(f,n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for (R,K,cols,cols2,ns,name) in ((384,16,4096,64,1,'time W'),(170,16,4096,64,3,'node W'),(384,4,1700,0,1,'logits'),(170,16,96,0,1,'A graph'),(32,4,1920,0,1,'dyn')):
    emb=torch.randn(R,K,device=dev); pool=torch.randn(K,cols,device=dev); pool2=torch.randn(K,cols2,device=dev) if cols2 else None
    dW=torch.randn(ns*R,cols,device=dev); dW2=torch.randn(R,cols2,device=dev) if cols2 else None
    dpool=torch.zeros(K,cols,device=dev); dpool2=torch.zeros(K,cols2,device=dev) if cols2 else None; demb=torch.zeros(R,K,device=dev)
    print(name, 'fwd %.1f us' % t(lambda: ops.poolgen(emb,pool,pool2)), 'bwd_pool %.1f us' % t(lambda: ops.poolgen_bwd_pool(emb,dW,dpool,dW2,dpool2,nsplit=ns)),
          'bwd_emb %.1f us' % t(lambda: ops.poolgen_bwd_emb(dW,pool,demb,dW2,pool2,nsplit=ns)))
x=torch.randn(6300000//4*4,device=dev)
print('copy 6.3MB %.1f us' % t(lambda: x.clone()), ' zeros %.1f us' % t(lambda: torch.zeros(384,64,device=dev)))
