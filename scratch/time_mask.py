import sys; sys.path.insert(0,'.')
import torch, time
from gptst_amd import ops, synth
dev='cuda:0'
M=65280
noise=synth.make_noise(M,1).to(dev); na=synth.make_noise(M,2).to(dev); nr=synth.make_noise(M,3).to(dev)
prob=torch.softmax(torch.randn(M,10,device=dev)*2,-1)
lc=torch.tensor(synth.class_order(10,5),dtype=torch.int32,device=dev); nums=torch.tensor([8160,8160],dtype=torch.int32,device=dev)
def t(f,n=50):
    f(); torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print('mask_random us', t(lambda: ops.mask_random(noise, M//4)))
lab,cnt=ops.mask_labels(prob)
print('mask_labels us', t(lambda: ops.mask_labels(prob)))
print('mask_adaptive us', t(lambda: ops.mask_adaptive(lab,cnt,lc,nums,na,nr,1,1)))
