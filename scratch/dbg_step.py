import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from golden_util import load, t
from gptst_amd import synth, ops, engine
from gptst_amd.config import make_args
from oracle import gptst_oracle as O
from gptst_amd.model import GPTST_Model
from gptst_amd.step import PretrainStep
DEV='cuda:0'
fx = load("steps.npz")
args = make_args("PEMS08", num_nodes=20, embed_dim=8, HS=5, HT=6, num_route=2, scaler_zeros=synth.scaler_zeros(), epochs=30, change_epoch=3)
sd = O.init_state_dict(args, int(fx["sd_seed"]))
model = GPTST_Model(args); model.load_state_dict(sd); model = model.to(DEV)
st = PretrainStep(model, args, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=4, use_graph=False)
for step in range(9):
    epoch = int(fx["st%d.epoch" % step]); tag="st%d"%step
    src = synth.make_batch(4, 12, 20, 1, seed=500 + step, start_slot=17 * step).to(DEV)
    if epoch <= args.change_epoch: st.step(src, epoch, noise=t(fx, tag + ".noise0").to(DEV))
    else:
        st.step(src, epoch, noise_a=t(fx, tag + ".noise0").to(DEV), noise_r=t(fx, tag + ".noise1").to(DEV), list_c=[int(i) for i in fx[tag + ".list_c"]])
    print(step, epoch, st.losses(), fx['losses'][step], 'masked', int((st.last_mask==0).sum()), 'ctrl', st.ctrl.cpu().tolist(), 'stats', st.stats_out.cpu().tolist()[:4], 'flat nan', bool(torch.isnan(model.flat).any()), 'hyper', st.hyper.cpu().tolist()[:4])
    if epoch > args.change_epoch:
        p = model.param_views()
        prob,_ = engine.guide_fwd(p, st.src, st.src[:,:,0,1:3].contiguous(), st.dims, 1)
        lab, cnt = ops.mask_labels(prob)
        print('   prob nan', bool(torch.isnan(prob).any()), 'counts', cnt.cpu().tolist())
