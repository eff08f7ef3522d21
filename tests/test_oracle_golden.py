"""Pins oracle/gptst_oracle.py against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerances: the oracle is op-for-op the same ATen
sequence, so forward is expected bit-equal; 1e-6 rel is allowed for thread-count dependent
reduction order."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, cfg_args, check, load, t
from gptst_amd import synth
from gptst_amd.config import make_args
from oracle import gptst_oracle as O

RT, AT = 2e-6, 1e-7


def _sd(fx, tag, names):
    return {n: t(fx, "%s.p.%s" % (tag, n)).clone().requires_grad_() for n in names}


def _grads(out, gout, leaves):
    (out * gout).sum().backward()
    return [x.grad for x in leaves]


def test_init_kat_matches_reference():
    kat = json.load(open(os.path.join(GOLDEN, "init_kat.json")))
    for ds, k in kat.items():
        args = make_args(ds)
        sd = O.init_state_dict(args, k["seed"])
        assert list(sd.keys()) == k["keys"]
        assert [list(v.shape) for v in sd.values()] == k["shapes"]
        assert O.state_hash(sd) == k["sha256"], ds
    # the survey's published KAT for PEMS08 seed 12
    assert kat["PEMS08"]["sha256"] == "306ea7aca00c80cdd4b7508dc28dd0ea2d3670fe29d3a5de9c0e5179325dae2d"
    assert kat["PEMS08"]["nparams"] == 1036531 and len(kat["PEMS08"]["keys"]) == 159


def test_squash():
    fx = load("modules.npz")
    check(fx, "squash.y", O.squash(t(fx, "squash.x")), RT, AT)
    assert torch.all(O.squash(torch.zeros(2, 8)) == 0)


@pytest.mark.parametrize("tag,fn", [("tf16", O.time_feature), ("tf4", O.time_feature), ("tfs", O.time_feature_spg)])
def test_time_feature(tag, fn):
    fx = load("modules.npz")
    names = [k.split(".p.")[1] for k in fx.files if k.startswith(tag + ".p.")]
    sd = _sd(fx, tag, names)
    inp = t(fx, tag + ".in").clone().requires_grad_()
    out = fn({"x." + k: v for k, v in sd.items()}, "x.", inp)
    check(fx, tag + ".out", out, RT, AT)
    _grads(out, t(fx, tag + ".gout"), [])
    check(fx, tag + ".gin", inp.grad, 1e-5, 1e-6)
    for n in names:
        check(fx, "%s.g.%s" % (tag, n), sd[n].grad, 1e-5, 1e-6)


@pytest.mark.parametrize("tag", ["ht_a", "ht_b"])
def test_hypertem(tag):
    fx = load("modules.npz")
    sd = _sd(fx, tag, ["adj", "weights_pool", "bias_pool"])
    x, ne, te = (t(fx, "%s.%s" % (tag, n)).clone().requires_grad_() for n in ("x", "ne", "te"))
    out = O.hypertem({"h." + k: v for k, v in sd.items()}, "h.", x, ne, te)
    check(fx, tag + ".out", out, RT, AT)
    _grads(out, t(fx, tag + ".gout"), [])
    for n, v in (("x", x), ("ne", ne), ("te", te)):
        check(fx, "%s.g.%s" % (tag, n), v.grad, 2e-5, 1e-5)
    for n, v in sd.items():
        check(fx, "%s.g.%s" % (tag, n), v.grad, 2e-5, 1e-5)


@pytest.mark.parametrize("tag", ["cap_a", "cap_b"])
@pytest.mark.parametrize("m5d", [True, False])
def test_cap(tag, m5d):
    fx = load("modules.npz")
    names = ["t_adj", "adj", "weights_spa", "bias_spa", "ln_p.weight", "ln_p.bias"]
    sd = _sd(fx, tag, names)
    full = {"c." + k: v for k, v in sd.items()}
    full["c.mask_template"] = t(fx, tag + ".mask_template")
    x, ne, tes, teb = (t(fx, "%s.%s" % (tag, n)).clone().requires_grad_() for n in ("x", "ne", "tes", "teb"))
    out, c, dyn = O.cap(full, "c.", x, ne, tes, teb, int(fx[tag + ".R"]), materialize_5d=m5d)
    tol = (RT, AT) if m5d else (2e-5, 2e-6)   # the fused-routing identity reorders fp32 sums
    check(fx, tag + ".out", out, *tol)
    check(fx, tag + ".c", c, *tol)
    check(fx, tag + ".dyn", dyn, RT, AT)
    _grads(out, t(fx, tag + ".gout"), [])
    for n, v in (("x", x), ("ne", ne), ("tes", tes), ("teb", teb)):
        check(fx, "%s.g.%s" % (tag, n), v.grad, 5e-5, 2e-5)
    for n, v in sd.items():
        check(fx, "%s.g.%s" % (tag, n), v.grad, 5e-5, 2e-5)


@pytest.mark.parametrize("tag", ["mlp_a", "mlp_b"])
def test_mlp_rl(tag):
    fx = load("modules.npz")
    names = ["weights_pool_spa", "bias_pool_spa", "weights_pool_tem", "bias_pool_tem", "ln1.weight", "ln1.bias",
             "ln3.weight", "ln3.bias"]
    sd = _sd(fx, tag, names)
    eb, te, ne = (t(fx, "%s.%s" % (tag, n)).clone().requires_grad_() for n in ("eb", "te", "ne"))
    out = O.mlp_rl({"m." + k: v for k, v in sd.items()}, "m.", eb, te, ne)
    check(fx, tag + ".out", out, RT, AT)
    _grads(out, t(fx, tag + ".gout"), [])
    for n, v in (("eb", eb), ("te", te), ("ne", ne)):
        check(fx, "%s.g.%s" % (tag, n), v.grad, 2e-5, 1e-5)
    for n, v in sd.items():
        check(fx, "%s.g.%s" % (tag, n), v.grad, 2e-5, 1e-5)


def _small_args(fx, tag):
    return cfg_args(fx, tag, make_args, scaler_zeros=synth.scaler_zeros())


def _inject(fx, tag, epoch, args):
    if epoch <= args.change_epoch:
        return dict(noise=t(fx, tag + ".noise0"))
    return dict(noise_a=t(fx, tag + ".noise0"), noise_r=t(fx, tag + ".noise1"), list_c=[int(i) for i in fx[tag + ".list_c"]])


@pytest.mark.parametrize("tag", ["s_rand", "s_ada_all", "s_ada_half", "s_ada_full", "s_base2"])
def test_small_forward_backward(tag):
    """Full model forward 5-tuple (mask bit-exact) + loss + every parameter gradient vs the reference."""
    fx = load("forward_small.npz")
    args = _small_args(fx, tag)
    epoch = int(fx[tag + ".epoch"])
    sd = O.init_state_dict(args, int(fx[tag + ".sd_seed"]))
    assert O.state_hash(sd) == str(fx[tag + ".sd_hash"])
    st = O.Stepper(sd, args, synth.SCALER_MEAN, synth.SCALER_STD)
    src = t(fx, tag + ".src")
    outs, aux = O.forward_pretrain(st.sd, args, src, epoch, **_inject(fx, tag, epoch, args))
    out, dec, mask, prob, hs1 = outs
    assert torch.equal(mask.to(torch.int8), t(fx, tag + ".mask")), "mask must be bit-exact"
    check(fx, tag + ".out", out, RT, AT)
    check(fx, tag + ".dec", dec, RT, AT)
    check(fx, tag + ".prob", prob, RT, AT)
    check(fx, tag + ".hs1", hs1, RT, AT)
    loss, lf, ls = O.pretrain_loss(outs, src, args, epoch, synth.SCALER_MEAN, synth.SCALER_STD)
    np.testing.assert_allclose([float(loss), float(lf), float(ls)], fx[tag + ".loss"], rtol=1e-6)
    loss.backward()
    for k, p in st.sd.items():
        if k.endswith("mask_template"):
            continue
        g = p.grad if p.grad is not None else torch.zeros(0)
        check(fx, "%s.grad.%s" % (tag, k), g, 1e-4, 1e-5, what=tag)


def test_full_forward_pems08():
    """PEMS08 dims, B=2, seed-12 init: 5-tuple at epochs 1/11/200/300 and the eval-mode embedding."""
    fx = load("forward_full.npz")
    args = make_args("PEMS08", scaler_zeros=synth.scaler_zeros())
    sd = O.init_state_dict(args, 12)
    src = t(fx, "src")
    torch.testing.assert_close(src, synth.make_batch(2, 12, 170, 1, seed=1234))
    for epoch in (1, 11, 200, 300):
        tag = "e%d" % epoch
        with torch.no_grad():
            (out, dec, mask, prob, hs1), _ = O.forward_pretrain(sd, args, src, epoch, **_inject(fx, tag, epoch, args))
        assert torch.equal(mask.to(torch.int8), t(fx, tag + ".mask")), epoch
        assert int(mask.sum()) == 1020
        check(fx, tag + ".out", out, RT, AT)
        check(fx, tag + ".prob", prob, RT, AT)
        check(fx, tag + ".hs1", hs1, RT, AT)
        check(fx, tag + ".dec_sub", dec[:, :, ::7, ::5], RT, AT)
        st = fx[tag + ".dec_stats"]
        assert abs(float(dec.double().abs().mean()) - st[1]) < 1e-6 * st[1]
    with torch.no_grad():
        emb = O.forward_eval(sd, args, src)
    check(fx, "eval.emb_sub", emb[:, :, ::7, ::5], RT, AT)


def test_step_sequence():
    """12 optimiser steps (6 random-phase, 6 adaptive+KL) reproduce the reference loss sequence, masks,
    final weights and the per-parameter Adam step counts (grad-None skipping)."""
    fx = load("steps.npz")
    args = make_args("PEMS08", num_nodes=20, embed_dim=8, HS=5, HT=6, num_route=2, scaler_zeros=synth.scaler_zeros(),
                     epochs=30, change_epoch=3)
    sd = O.init_state_dict(args, int(fx["sd_seed"]))
    assert O.state_hash(sd) == str(fx["sd0_hash"])
    st = O.Stepper(sd, args, synth.SCALER_MEAN, synth.SCALER_STD)
    losses = fx["losses"]
    for step in range(losses.shape[0]):
        epoch = int(fx["st%d.epoch" % step])
        src = synth.make_batch(4, 12, 20, 1, seed=500 + step, start_slot=17 * step)
        loss, lf, ls, outs, aux = st.step(src, epoch, **_inject(fx, "st%d" % step, epoch, args))
        assert torch.equal(outs[2].to(torch.int8), t(fx, "st%d.mask" % step)), step
        # fp32 round-off is amplified by the training dynamics after ~10 steps (observed 1e-7 -> 3e-4 on
        # the KL term); the north-star loss-curve tolerance is 1e-3.
        np.testing.assert_allclose([loss, lf, ls], losses[step], rtol=1e-5 if step < 10 else 1e-3, err_msg="step %d" % step)
    # Final weights: Adam turns a sign flip of a round-off-level gradient into a +-lr move, so individual
    # elements are chaotic; the tensor-level relative L2 error still pins clip + Adam (a wrong optimiser
    # moves every element by ~steps*lr = 3.6e-2, i.e. rel-L2 >= 1e-1).
    for k, v in st.sd.items():
        key = "sdN." + k
        if key in fx.files:
            ref, val = t(fx, key).reshape(-1), v.detach().reshape(-1)
        else:
            ref, val = t(fx, key + "::sub"), v.detach().reshape(-1)[::13]
        rel = float((val - ref).norm() / ref.norm().clamp_min(1e-12))
        assert rel < 2e-2, (k, rel)
    steps = [int(st.opt.state[p]["step"]) if p in st.opt.state else 0 for p in st.params]
    assert steps == [int(s) for s in fx["adam_steps"]]


def test_trainer_checkpoint_was_accepted_by_the_reference():
    """Drop-in, consumer side (reference model/Model.py:95-98: torch.load -> GPTST_Model.load_state_dict, strict): the checkpoint FILE the product's
    trainer wrote on the MI355X (tests/golden/trainer_ckpt.pth, by tests/golden/make_trainer_ckpt.py) was loaded into the REFERENCE module in
    the build container by tests/golden/check_ckpt_in_reference.py, whose record is committed next to it.  Here: the record belongs to this
    very file, says strict load + both forwards agree with the oracle, and the file loads into the product's module and the oracle alike."""
    import hashlib
    import json
    import os
    import torch
    from gptst_amd.config import make_args
    from gptst_amd.model import GPTST_Model
    from gptst_amd import synth
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rec = json.load(open(os.path.join(here, "ckpt_in_reference.json")))
    meta = json.load(open(os.path.join(here, "trainer_ckpt.json")))
    path = os.path.join(here, "trainer_ckpt.pth")
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == rec["sha256"], "re-run tests/golden/check_ckpt_in_reference.py on the committed file"
    assert rec["strict_load"] == "ok" and rec["keys"] == 159
    assert all(v < 2e-6 for v in rec["forward_max_rel_err_vs_oracle"].values())
    sd = torch.load(path, map_location="cpu")
    args = make_args("PEMS08", scaler_zeros=meta["scaler_zeros"], **meta["args"])
    assert list(sd.keys()) == list(O.init_state_dict(args, 1).keys())
    m = GPTST_Model(args)
    m.load_state_dict(sd, strict=True)
    emb = O.forward_eval(sd, args, synth.make_batch(2, 12, meta["args"]["num_nodes"], 1, seed=1))
    assert torch.isfinite(emb).all() and float(emb.abs().max()) > 0
    assert meta["steps"] > 100                                   # a trained file, not an initial state
