"""GPU parity of the whole model (forward 5-tuple, loss, every parameter gradient) against the pinned oracle at the shapes of the
other BASELINE configs: METR_LA (N=207, ada_type 'half'; HS*N not a multiple of 4 -> scalar column kernels), NYC_TAXI (N=266,
2 channels, mape_thresh 0.001), cluster-count sweep HS in {2,5,20,40}, deeper routing."""
import pytest
import torch

from gptst_amd import synth
from gptst_amd.config import make_args
from oracle import gptst_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = {
    "metr_la": dict(ds="METR_LA", over={}, B=2, epoch=200),
    "metr_la_rand": dict(ds="METR_LA", over={}, B=2, epoch=3),
    "nyc_taxi": dict(ds="NYC_TAXI", over={}, B=2, epoch=150),
    "nyc_bike_rand": dict(ds="NYC_BIKE", over={}, B=1, epoch=1),
    "hs2": dict(ds="NYC_TAXI", over=dict(HS=2, num_nodes=61), B=2, epoch=100),
    "hs5": dict(ds="PEMS08", over=dict(HS=5, num_nodes=50), B=2, epoch=100),
    "hs20": dict(ds="PEMS08", over=dict(HS=20, num_nodes=45), B=2, epoch=100),
    "hs40": dict(ds="NYC_TAXI", over=dict(HS=40, num_nodes=37), B=1, epoch=100),
    # BASELINE configs[3] at its real node count: the (b,t) capsule matrix fits LDS for some cap kernels and not for others, so one
    # cap mixes LDS and streaming kernels (per-kernel ESHAPE fallback, ops._lds_or_stream); HS = 40 also takes the global cap_cross path
    "nyc266_hs20": dict(ds="NYC_TAXI", over=dict(HS=20), B=1, epoch=100),
    "nyc266_hs40": dict(ds="NYC_TAXI", over=dict(HS=40), B=1, epoch=100),
    "nyc266_hs20_rand": dict(ds="NYC_TAXI", over=dict(HS=20), B=1, epoch=2),
    "route4": dict(ds="PEMS08", over=dict(num_route=4, num_nodes=33, embed_dim=8), B=2, epoch=100),
    "c128": dict(ds="PEMS08", over=dict(hidden_dim=128, num_nodes=40, embed_dim=8), B=2, epoch=100),
    "c128_rand": dict(ds="NYC_TAXI", over=dict(hidden_dim=128, num_nodes=23, embed_dim=4), B=1, epoch=2),
    "n600": dict(ds="PEMS08", over=dict(num_nodes=600, embed_dim=8), B=1, epoch=100),                 # capsule matrix beyond LDS: capbig path
    "n260_c128": dict(ds="PEMS08", over=dict(num_nodes=260, hidden_dim=128, embed_dim=8), B=1, epoch=100),   # config-5 style (C = 128)
    # BASELINE configs[4] at its FULL size (N = 4096, C = 128, d = 16, HS = 10; one sample): streaming cap, apply128 / wgrad128 / tmix
    # kernels against the oracle's forward, loss and every gradient (the oracle needs ~10 s of CPU for it with materialize_5d=False)
    # Reference in fp64: at this size the fp32 ORACLE's own gradient of encoder.neb4mask is 1.2e-3 off its fp64 run (4096-node sums).
    "c5_full_n4096_c128": dict(ds="PEMS08", over=dict(num_nodes=4096, hidden_dim=128), B=1, epoch=100, f64=True),
}


# Whole-model gradient bound = the north star's 1e-4 (r04: 2e-4).  Measured worst per case: profiles/parity_r05.json.
GRAD_TOL = 1e-4
# Parameters allowed past it, per case, under the fp64 rule at the end of the test.  hs5 (PEMS08-style base = 1, N = 50, HS = 5): the gradient of
# the cap's time embedding `teb` (GPTST.py:104,260) is the sum over (h, n) of softmax-backward terms that cancel to 1e-5 of their size — |g| ~ 6e-6
# where the node embeddings have 5e-2 — and the fp32 ORACLE is itself 6.9e-5 off its fp64 run there (HIP: 8.6e-5; every round-4 kernel switch
# leaves the figure unchanged to three digits — it is the conditioning of the quantity, not a kernel).
ILL_CONDITIONED = {"hs5": ("encoder.STHCN_encode.time_feature1_.",)}


@pytest.mark.parametrize("name", list(CASES))
def test_model_vs_oracle(name, parity):
    from gptst_amd.model import GPTST_Model
    c = CASES[name]
    args = make_args(c["ds"], scaler_zeros=synth.scaler_zeros(), **c["over"])
    B, T, N, base, HS = c["B"], 12, args.num_nodes, args.input_base_dim, args.HS
    sd = O.init_state_dict(args, 11)
    src = synth.make_batch(B, T, N, base, interval=args.interval, seed=21)
    M = B * T * N
    epoch = c["epoch"]
    if epoch <= args.change_epoch:
        inj = dict(noise=synth.make_noise(M * base, 5))
    else:
        inj = dict(noise_a=synth.make_noise(M, 5), noise_r=synth.make_noise(M, 6), list_c=synth.class_order(HS, 3))
    import os
    rdt = torch.float64 if (c.get("f64") or os.environ.get("GPTST_TEST_F64") == "1") else torch.float32
    cast = lambda v: v.to(rdt) if torch.is_tensor(v) and v.dtype.is_floating_point else v      # noqa: E731
    st = O.Stepper({k: cast(v) for k, v in sd.items()}, args, synth.SCALER_MEAN, synth.SCALER_STD, materialize_5d=False)
    outs_r, aux = O.forward_pretrain(st.sd, args, cast(src), epoch, materialize_5d=False, **{k: cast(v) for k, v in inj.items()})
    loss_r, lf_r, ls_r = O.pretrain_loss(outs_r, cast(src), args, epoch, synth.SCALER_MEAN, synth.SCALER_STD)
    loss_r.backward()
    outs_r = tuple(o.float() if torch.is_tensor(o) and o.dtype == torch.float64 else o for o in outs_r)

    model = GPTST_Model(args)
    model.load_state_dict(sd)
    model = model.to(DEV)
    model.set_mask_inputs(**inj)
    srcd = src.to(DEV)
    out, dec, mask, prob, hs1 = model(srcd, srcd, None, epoch)
    # masks: bit-exact unless an fp32-level argmax flip changes a label (adaptive phase); then teacher-force
    same = torch.equal(mask.cpu(), outs_r[2])
    if not same:
        agree = float((mask.cpu() == outs_r[2]).float().mean())
        assert epoch > args.change_epoch and agree > 0.97, (name, agree)
        model.set_mask_inputs(forced_mask=aux["final_mask"].float())
        out, dec, mask, prob, hs1 = model(srcd, srcd, None, epoch)
        assert torch.equal(mask.cpu(), outs_r[2])

    def rel(a, b):
        a, b = a.detach().cpu().double(), b.detach().double()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))

    for nm, a_, b_ in (("out", out, outs_r[0]), ("dec", dec, outs_r[1]), ("prob", prob, outs_r[3]), ("hs1", hs1, outs_r[4])):
        e = rel(a_, b_)
        parity("fwd_" + nm, e)
        assert e < 1e-5, (nm, e)            # measured <= 1.3e-6
    p = (out * synth.SCALER_STD + synth.SCALER_MEAN) * mask
    y = (srcd[..., :base] * synth.SCALER_STD + synth.SCALER_MEAN) * mask
    keep = y > args.mape_thresh
    loss = torch.abs(torch.masked_select(y, keep) - torch.masked_select(p, keep)).mean()
    if epoch > args.change_epoch:
        loss = loss + torch.nn.functional.kl_div(prob.log(), hs1, reduction="sum") * 0.1
    parity("loss", abs(float(loss) - float(loss_r)) / abs(float(loss_r)))
    assert abs(float(loss) - float(loss_r)) < 2e-6 * abs(float(loss_r))      # measured <= 1.5e-7
    loss.backward()
    worst, over = 0.0, {}
    for k, pm in model.named_parameters():
        gr = st.sd[k].grad
        if gr is None:
            assert pm.grad is None or float(pm.grad.abs().max()) == 0.0, k
            continue
        e = rel(pm.grad, gr)
        worst = max(worst, e)
        if e >= GRAD_TOL:
            over[k] = e
    parity("grad_worst", worst)
    print(name, "worst grad rel err %.2e" % worst)
    if over:
        # A parameter past 1e-4 must be a NAMED ill-conditioned one, and then the yardstick is the fp64 oracle: the HIP gradient may be no
        # further from it than 1.5x the fp32 ORACLE's own distance (tools/grad_bisect.py, profiles/r05_grad_bisect.txt: for these tensors the
        # two fp32 implementations err by the same amount in opposite directions, so their mutual distance is the sum)
        assert rdt == torch.float32
        assert all(any(k.startswith(pfx) for pfx in ILL_CONDITIONED.get(name, ())) for k in over), (name, over)
        c64 = lambda v: v.double() if torch.is_tensor(v) and v.dtype.is_floating_point else v      # noqa: E731
        st64 = O.Stepper({k: c64(v) for k, v in sd.items()}, args, synth.SCALER_MEAN, synth.SCALER_STD, materialize_5d=False)
        o64, aux64 = O.forward_pretrain(st64.sd, args, c64(src), epoch, materialize_5d=False, **{k: c64(v) for k, v in inj.items()})
        assert torch.equal(aux64["final_mask"], aux["final_mask"])
        O.pretrain_loss(o64, c64(src), args, epoch, synth.SCALER_MEAN, synth.SCALER_STD)[0].backward()
        for k in over:
            g64 = st64.sd[k].grad
            e_hip, e_orc = rel(dict(model.named_parameters())[k].grad, g64), rel(st.sd[k].grad, g64)
            parity("grad_vs_f64:" + k, e_hip)
            parity("oracle_vs_f64:" + k, e_orc)
            assert e_hip < GRAD_TOL and e_hip <= 1.5 * e_orc, (name, k, e_hip, e_orc)
