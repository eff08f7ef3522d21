import sys; sys.path.insert(0,'.')
import torch
from gptst_amd import ops, _C
dev='cuda:0'
def run(f,n=20):
    for _ in range(n): f()
    torch.cuda.synchronize()
for nch in (2,4,8):
    _C.lib().value('gptst_tune', 1, nch)
    for (R,K,cols,cols2,ns,name) in ((384,16,4096,64,1,'time W'),(170,16,4096,64,3,'node W')):
        emb=torch.randn(R,K,device=dev); pool=torch.randn(K,cols,device=dev); pool2=torch.randn(K,cols2,device=dev)
        dW=torch.randn(ns*R,cols,device=dev); dW2=torch.randn(R,cols2,device=dev)
        dpool=torch.zeros(K,cols,device=dev); dpool2=torch.zeros(K,cols2,device=dev); demb=torch.zeros(R,K,device=dev)
        run(lambda: ops.poolgen_bwd_pool(emb,dW,dpool,dW2,dpool2,nsplit=ns))
        if nch==2: run(lambda: ops.poolgen_bwd_emb(dW,pool,demb,dW2,pool2,nsplit=ns))
for (R,K,cols,ns) in ((384,4,1700,1),(170,16,96,1),(32,4,1920,1),(384,4,2070,1)):
    emb=torch.randn(R,K,device=dev); pool=torch.randn(K,cols,device=dev); dW=torch.randn(ns*R,cols,device=dev)
    dpool=torch.zeros(K,cols,device=dev); demb=torch.zeros(R,K,device=dev)
    run(lambda: ops.poolgen_bwd_pool(emb,dW,dpool,nsplit=ns)); run(lambda: ops.poolgen_bwd_emb(dW,pool,demb,nsplit=ns))
