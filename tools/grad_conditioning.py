"""Where the whole-model gradient of `time_feature1_` loses its digits, on the CPU oracle alone: fp32 vs fp64 runs of one
case of tests/test_gpu_shapes.py, stage by stage — the cluster-logit gradient dl, its contraction d_teb (GPTST.py:104), then the parameter
gradients of the time-feature MLP (GPTST.py:198-202).  Result (profiles/r05_grad_bisect.txt): dl and d_teb agree to 1e-6; the loss of four
digits happens INSIDE the MLP's backward between ln2 and ln1, where W_ln2^T dh2 cancels to 1 % of its terms — any fp32 upstream rounding of 1e-6
is amplified to 1e-4 there, in the oracle as in the HIP path.

    python tools/grad_conditioning.py hs5
"""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gptst_amd import synth
from gptst_amd.config import make_args
from oracle import gptst_oracle as O
from test_gpu_shapes import CASES
name=sys.argv[1]
c=CASES[name]
args = make_args(c["ds"], scaler_zeros=synth.scaler_zeros(), **c["over"])
B,T,N,base,HS=c["B"],12,args.num_nodes,args.input_base_dim,args.HS
sd=O.init_state_dict(args,11)
src=synth.make_batch(B,T,N,base,interval=args.interval,seed=21)
M=B*T*N
inj=dict(noise_a=synth.make_noise(M,5),noise_r=synth.make_noise(M,6),list_c=synth.class_order(HS,3))
orig_einsum=torch.einsum
def run(dt):
    keep=[]
    def ein(eq,*ops):
        r=orig_einsum(eq,*ops)
        if eq=="btd,dhn->bthn":
            r.retain_grad(); keep.append((r,ops[0],ops[1]))
        return r
    torch.einsum=ein
    cast=lambda v: v.to(dt) if torch.is_tensor(v) and v.dtype.is_floating_point else v
    st=O.Stepper({k:cast(v) for k,v in sd.items()},args,synth.SCALER_MEAN,synth.SCALER_STD,materialize_5d=False)
    outs,aux=O.forward_pretrain(st.sd,args,cast(src),c["epoch"],materialize_5d=False,**{k:cast(v) for k,v in inj.items()})
    loss,_,_=O.pretrain_loss(outs,cast(src),args,c["epoch"],synth.SCALER_MEAN,synth.SCALER_STD)
    loss.backward()
    torch.einsum=orig_einsum
    return keep, st
k32,st32=run(torch.float32); k64,st64=run(torch.float64)
rel=lambda a,b: float((a.double()-b.double()).abs().max()/b.double().abs().max())
for i,((r32,t32,a32),(r64,t64,a64)) in enumerate(zip(k32,k64)):
    dl32,dl64=r32.grad,r64.grad
    dteb64=orig_einsum("bthn,dhn->btd",dl64,a64)
    dteb32=orig_einsum("bthn,dhn->btd",dl32,a32)
    dteb_mixed=orig_einsum("bthn,dhn->btd",dl32.double(),a32.double())
    terms=orig_einsum("bthn,dhn->btd",dl64.abs(),a64.abs())
    print(i,"dl rel err %.2e"%rel(dl32,dl64),"dteb f32 %.2e"%rel(dteb32,dteb64),"dteb(f32 dl, f64 contraction) %.2e"%rel(dteb_mixed,dteb64),
          "cancellation |sum|/sum|.| %.2e"%float(dteb64.abs().max()/terms.max()), "sum_h dl max %.2e vs |dl| %.2e"%(float(dl64.sum(2).abs().max()),float(dl64.abs().max())))
# total d_teb per STHCN (sum of both caps) and the parameter gradients
for pfx in ("encoder.STHCN_encode.","decoder.STHCN_decode."):
    for nm in ("ln.weight","ln.bias","ln2.weight","ln2.bias","ln1.weight","ln1.bias","ln_day.weight","ln_day.bias","ln_week.weight"):
        k=pfx+"time_feature1_."+nm
        g32,g64=st32.sd[k].grad,st64.sd[k].grad
        print("%-55s f32-f64 %.2e |g|max %.2e"%(k,rel(g32,g64),float(g64.abs().max())))
# tensor-level: teb.grad
for i in (0,2):
    t32,t64=k32[i][1],k64[i][1]
    print("teb grad (both caps) err", rel(t32.grad,t64.grad) if t32.grad is not None else None)
