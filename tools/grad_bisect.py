"""Locate which engine path loses gradient accuracy on a whole-model case.

For each case of tests/test_gpu_shapes.py::CASES named on the command line, the oracle runs once in fp32 and once in fp64; the HIP model
runs under the default engine flags and with each round-4 switch off in turn.  Per configuration the script prints the worst
parameters: HIP-vs-fp64 error, the fp32 oracle's own distance to fp64 (the yardstick), and HIP-vs-fp32-oracle (what the test asserts).

    python tools/grad_bisect.py hs5 hs2 nyc_taxi > gpurun_out/grad_bisect.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from gptst_amd import engine, synth, step as step_mod      # noqa: E402
from gptst_amd.config import make_args                     # noqa: E402
from oracle import gptst_oracle as O                       # noqa: E402
from test_gpu_shapes import CASES                          # noqa: E402

DEV = "cuda:0"
SWITCHES = [("default", {})] if os.environ.get("BISECT_DEFAULT_ONLY") == "1" else [("default", {}), ("ENCIN=0", dict(ENCIN=False)), ("GUIDEIN=0", dict(GUIDEIN=False)), ("PAIR_BWD=0", dict(PAIR_BWD=False)),
            ("CHAIN_FWD=0", dict(CHAIN_FWD=False)), ("CROSS_ROLE=0", dict(CROSS_ROLE=0)), ("FUSE_CROSS=0", dict(FUSE_CROSS=False)),
            ("all r04 off", dict(ENCIN=False, GUIDEIN=False, PAIR_BWD=False, CHAIN_FWD=False, CROSS_ROLE=0))]


def oracle_grads(c, args, sd, src, inj, dt):
    cast = lambda v: v.to(dt) if torch.is_tensor(v) and v.dtype.is_floating_point else v      # noqa: E731
    st = O.Stepper({k: cast(v) for k, v in sd.items()}, args, synth.SCALER_MEAN, synth.SCALER_STD, materialize_5d=False)
    outs, aux = O.forward_pretrain(st.sd, args, cast(src), c["epoch"], materialize_5d=False, **{k: cast(v) for k, v in inj.items()})
    loss, _, _ = O.pretrain_loss(outs, cast(src), args, c["epoch"], synth.SCALER_MEAN, synth.SCALER_STD)
    loss.backward()
    return {k: v.grad.double() for k, v in st.sd.items() if getattr(v, "grad", None) is not None}, aux, outs


def hip_grads(c, args, sd, src, inj, aux):
    from gptst_amd.model import GPTST_Model
    base = args.input_base_dim
    model = GPTST_Model(args)
    model.load_state_dict(sd)
    model = model.to(DEV)
    model.set_mask_inputs(forced_mask=aux["final_mask"].float(), **inj)
    srcd = src.to(DEV)
    out, dec, mask, prob, hs1 = model(srcd, srcd, None, c["epoch"])
    p = (out * synth.SCALER_STD + synth.SCALER_MEAN) * mask
    y = (srcd[..., :base] * synth.SCALER_STD + synth.SCALER_MEAN) * mask
    keep = y > args.mape_thresh
    loss = torch.abs(torch.masked_select(y, keep) - torch.masked_select(p, keep)).mean()
    if c["epoch"] > args.change_epoch:
        loss = loss + torch.nn.functional.kl_div(prob.log(), hs1, reduction="sum") * 0.1
    loss.backward()
    return {k: pm.grad.detach().cpu().double() for k, pm in model.named_parameters() if pm.grad is not None}


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def main():
    names = sys.argv[1:] or ["hs5", "hs2", "nyc_taxi"]
    for name in names:
        c = CASES[name]
        args = make_args(c["ds"], scaler_zeros=synth.scaler_zeros(), **c["over"])
        B, T, N, base, HS = c["B"], 12, args.num_nodes, args.input_base_dim, args.HS
        sd = O.init_state_dict(args, 11)
        src = synth.make_batch(B, T, N, base, interval=args.interval, seed=21)
        M = B * T * N
        if c["epoch"] <= args.change_epoch:
            inj = dict(noise=synth.make_noise(M * base, 5))
        else:
            inj = dict(noise_a=synth.make_noise(M, 5), noise_r=synth.make_noise(M, 6), list_c=synth.class_order(HS, 3))
        g32, aux, _ = oracle_grads(c, args, sd, src, inj, torch.float32)
        g64, aux64, _ = oracle_grads(c, args, sd, src, inj, torch.float64)
        print("==== %s  N=%d HS=%d base=%d B=%d epoch=%d   masks f32==f64: %s" % (
            name, N, HS, base, B, c["epoch"], bool(torch.equal(aux["final_mask"], aux64["final_mask"]))))
        yard = {k: rel(g32[k], g64[k]) for k in g64}
        print("fp32 oracle vs fp64 oracle: worst %.2e  (%s)" % (max(yard.values()), max(yard, key=yard.get)))
        saved = {k: getattr(engine, k) for _, d in SWITCHES for k in d}
        for label, d in SWITCHES:
            for k, v in saved.items():
                setattr(engine, k, v)
            for k, v in d.items():
                setattr(engine, k, v)
            try:
                gh = hip_grads(c, args, sd, src, inj, aux)
            except Exception as e:       # noqa: BLE001
                print("  [%s] FAILED %r" % (label, e))
                continue
            e64 = {k: rel(gh[k], g64[k]) for k in g64 if k in gh}
            e32 = {k: rel(gh[k], g32[k]) for k in g64 if k in gh}
            top = sorted(e32, key=e32.get, reverse=True)[:4]
            print("  [%-12s] worst vs f32-oracle %.2e  vs f64 %.2e" % (label, max(e32.values()), max(e64.values())))
            for k in top:
                print("        %-44s hip-f32o %.2e  hip-f64 %.2e  f32o-f64 %.2e  |g|max %.3e" % (
                    k, e32[k], e64[k], yard[k], float(g64[k].abs().max())))
        for k, v in saved.items():
            setattr(engine, k, v)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
