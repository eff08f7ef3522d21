import sys, torch
import os; R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R_); sys.path.insert(0,os.path.join(R_,'tests'))
from gptst_amd import synth, engine, ops
from gptst_amd.config import make_args
from gptst_amd.model import GPTST_Model
from gptst_amd.step import PretrainStep
from oracle import gptst_oracle as O
DEV='cuda:0'
args = make_args("PEMS08", num_nodes=20, embed_dim=8, HS=5, HT=6, num_route=2, scaler_zeros=synth.scaler_zeros(), epochs=30, change_epoch=3)
sd = O.init_state_dict(args, 11)
B=4; M=B*12*20
src = synth.make_batch(B,12,20,1,seed=700).to(DEV)
rec={}
o_tf=ops.timefeat_jobs_bwd; o_lin=ops.cap_cross_route_lin_bwd; o_rt=ops.cap_cross_route_bwd; o_lb=ops.linear_bwd
cur=None
def tf(tfl, tidx):
    torch.cuda.synchronize()
    rec[cur]['tf']=[t[2].clone() for t in tfl]
    return o_tf(tfl, tidx)
def lin(*a, **k):
    r=o_lin(*a, **k)
    if r is not None:
        rec[cur].setdefault('cap',[]).append(dict(dx=r[0].clone(), dWp=r[1].sum(0), dbp=r[2].sum(0), dl=r[3].clone(), ddyn=r[4].clone()))
    return r
def rt(*a, **k):
    r=o_rt(*a, **k)
    if r is not None: rec[cur].setdefault('cap',[]).append(dict(dl=r[1].clone(), ddyn=r[2].clone()))
    return r
def lb(*a, **k):
    r=o_lb(*a, **k)
    rec[cur]['cap'][-1].update(dx=r[0].clone(), dWp=r[1].sum(0), dbp=r[2].sum(0))
    return r
ops.timefeat_jobs_bwd=tf; ops.cap_cross_route_lin_bwd=lin; ops.cap_cross_route_bwd=rt; ops.linear_bwd=lb
for l in (False, True):
    cur=l; rec[l]={}
    engine.CAP_LIN = l
    model = GPTST_Model(args); model.load_state_dict(sd); model = model.to(DEV)
    st = PretrainStep(model, args, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=False, deterministic=True)
    st.step(src, 20, noise_a=synth.make_noise(M,70).to(DEV), noise_r=synth.make_noise(M,170).to(DEV), list_c=synth.class_order(5,7))
    torch.cuda.synchronize()
a,b=rec[False],rec[True]
print("caps recorded", len(a['cap']), len(b['cap']))
for i,(x,y) in enumerate(zip(a['cap'],b['cap'])):
    for k in ('dl','ddyn','dx','dWp','dbp'):
        d=float((x[k]-y[k]).abs().max()); s=float(x[k].abs().max())
        print(" cap %d %-5s maxdiff %.3e scale %.3e"%(i,k,d,s))
for i,(x,y) in enumerate(zip(a['tf'],b['tf'])):
    print(" timefeat input %d maxdiff %.3e scale %.3e"%(i,float((x-y).abs().max()),float(x.abs().max())))
