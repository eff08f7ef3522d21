"""GPU: the drop-in GPTST_Model under the reference's own training loop with its forward / backward as hipGraph replays (module_graph.py, r06) and
the fused clip + Adam behind the torch.optim interface (optim.ClipAdam).  The graphed node runs the fused stepper's kernels (low-rank first layers,
forward chains, dPre-chain backward); the eager node — the one tests/test_gpu_model.py pins against the REFERENCE's own outputs and gradients — is
the yardstick here: same source, the mask the graphed call drew teacher-forced into the eager call, same torch loss."""
import copy

import pytest
import torch

from gptst_amd import synth
from gptst_amd.config import make_args
from oracle import gptst_oracle as O
from test_gpu_model import _build, _loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _args(**over):
    kw = dict(num_nodes=20, embed_dim=8, HS=5, HT=6, num_route=2, scaler_zeros=synth.scaler_zeros(), epochs=30, change_epoch=3)
    kw.update(over)
    return make_args("PEMS08", **kw)


def _grads(model):
    return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


@pytest.mark.parametrize("shape", ["small", "bench"])
@pytest.mark.parametrize("epoch", [2, 20])
def test_graphed_node_equals_the_eager_node(epoch, shape, parity):
    from gptst_amd import module_graph
    args = _args() if shape == "small" else make_args("PEMS08", scaler_zeros=synth.scaler_zeros())
    if shape == "bench":
        epoch = 1 if epoch == 2 else 200
    B = 4 if shape == "small" else 32
    model = _build(args, O.init_state_dict(args, 5))
    assert module_graph.ENABLED
    for rep in range(3):                                   # capture, then two replays on other batches
        src = synth.make_batch(B, 12, args.num_nodes, 1, seed=40 + rep).to(DEV)
        model.zero_grad()
        outs = model(src, src, None, epoch)
        gp = next(iter(module_graph._CACHE[model].values()))
        assert gp.busy and gp.gen_id == rep + 1, "the training-loop call must take the graphed node"
        loss, lf, ls = _loss(outs, src, args, epoch)
        loss.backward()
        assert not gp.busy
        first = min(model._offs, key=model._offs.get)
        assert dict(model.named_parameters())[first].grad.data_ptr() == gp.gflat.data_ptr(), "autograd must keep the flat views (no per-tensor copy)"
        g_graph = _grads(model)
        mask_vis = (1 - outs[2]).float()
        # the eager node on the same source with that mask teacher-forced
        model.zero_grad()
        model.set_mask_inputs(forced_mask=mask_vis)
        ref = model(src, src, None, epoch)
        assert gp.gen_id == rep + 1, "injected mask inputs run eagerly"
        assert torch.equal(ref[2], outs[2])
        for a, b, nm, tol in zip(outs, ref, ("out", "dec", "mask", "prob", "hs1"), (2e-5, 2e-5, 0, 2e-5, 2e-5)):
            err = float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-6)
            parity("fwd_%s" % nm, err)
            assert err <= tol, (nm, err)
        loss2, _, _ = _loss(ref, src, args, epoch)
        assert abs(float(loss2) - float(loss)) <= 2e-5 * abs(float(loss2))
        loss2.backward()
        worst = 0.0
        for k, g_ref in _grads(model).items():
            g = g_graph[k]
            assert (g is None) == (g_ref is None), k
            if g is None:
                continue
            err = float((g - g_ref).abs().max()) / max(float(g_ref.abs().max()), 1e-6)
            worst = max(worst, err)
            assert err < 1e-4, (k, err)
        parity("grad_worst_graph_vs_eager", worst)


def test_calling_patterns_the_static_buffers_cannot_serve_fall_back():
    """no_grad, a second forward before the first one's backward, gradients that accumulate across backward calls: all correct, eagerly or by
    detaching the .grad tensors from the static buffer first."""
    from gptst_amd import module_graph
    args = _args()
    model = _build(args, O.init_state_dict(args, 6))
    src = synth.make_batch(4, 12, 20, 1, seed=50).to(DEV)
    with torch.no_grad():
        model(src, src, None, 2)
    assert model not in module_graph._CACHE or not module_graph._CACHE[model]
    o1 = model(src, src, None, 2)
    gp = next(iter(module_graph._CACHE[model].values()))
    o2 = model(src, src, None, 2)                          # o1's backward has not run: eager
    assert gp.gen_id == 1 and gp.busy
    (_loss(o1, src, args, 2)[0] + _loss(o2, src, args, 2)[0]).backward()
    assert not gp.busy
    g_sum = _grads(model)
    # accumulation: two graphed backward calls without zero_grad in between == the sum of the two single gradients
    model.zero_grad()
    masks, singles = [], []
    for i in range(2):
        model.zero_grad()
        o = model(src, src, None, 2)
        masks.append((1 - o[2]).float())
        _loss(o, src, args, 2)[0].backward()
        singles.append(_grads(model))
    model.zero_grad()
    for i in range(2):
        model.set_mask_inputs(forced_mask=masks[i])        # (eager, same masks) ...
        _loss(model(src, src, None, 2), src, args, 2)[0].backward()
    acc_ref = _grads(model)
    for k in acc_ref:
        if acc_ref[k] is None:
            continue
        want = singles[0][k] + singles[1][k]
        assert float((acc_ref[k] - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1e-6), k
    # ... and graphed with live .grad tensors: set_to_none=False keeps the views of the static buffer alive across the next replay
    model.zero_grad(set_to_none=False)
    o = model(src, src, None, 2)
    _loss(o, src, args, 2)[0].backward()
    ga = _grads(model)
    o = model(src, src, None, 2)
    assert gp.gen_id >= 4
    _loss(o, src, args, 2)[0].backward()
    gb = _grads(model)
    k0 = "encoder.STHCN_encode.cap1.ln_p.weight"
    assert float((gb[k0] - ga[k0]).abs().max()) > 0 and torch.isfinite(gb[k0]).all()
    # flow_decode in the loss: the graphed forward's backward is enqueued eagerly with the extra term == the eager node with the same mask
    model.zero_grad()
    o = model(src, src, None, 2)
    (_loss(o, src, args, 2)[0] + o[1].square().mean()).backward()
    gd = _grads(model)
    model.zero_grad()
    model.set_mask_inputs(forced_mask=(1 - o[2]).float())
    o2 = model(src, src, None, 2)
    (_loss(o2, src, args, 2)[0] + o2[1].square().mean()).backward()
    for k, g_ref in _grads(model).items():
        if g_ref is not None:
            assert float((gd[k] - g_ref).abs().max()) <= 1e-4 * max(float(g_ref.abs().max()), 1e-6), k
    copy.deepcopy(model)                                   # BasicTrainer.py:180 deep-copies the model: the graphs live outside it
    assert g_sum[k0] is not None


def test_clip_adam_optimizer_equals_clip_grad_norm_plus_torch_adam(parity):
    """gptst_amd.optim.ClipAdam == torch.nn.utils.clip_grad_norm_(5) + torch.optim.Adam over six steps of the reference loop that cross the phase
    switch (KL-path parameters get their first gradients — and their own Adam step count — at step 4), with a learning-rate change on the way."""
    from gptst_amd.optim import ClipAdam
    args = _args()
    sd = O.init_state_dict(args, 7)
    ma, mb = _build(args, sd), _build(args, sd)
    oa = torch.optim.Adam(ma.parameters(), lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False)
    ob = ClipAdam(mb.parameters(), lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False, max_grad_norm=args.max_grad_norm)
    for step in range(6):
        epoch = 2 if step < 3 else 20
        src = synth.make_batch(4, 12, 20, 1, seed=70 + step).to(DEV)
        oa.zero_grad(); ob.zero_grad()
        o = ma(src, src, None, epoch)
        _loss(o, src, args, epoch)[0].backward()
        raw = {k: (None if q.grad is None else q.grad.detach().clone()) for k, q in ma.named_parameters()}
        norm = torch.nn.utils.clip_grad_norm_(ma.parameters(), args.max_grad_norm)
        oa.step()
        # model B must see EXACTLY model A's gradients (Adam normalises per element: two fp32 backward passes differ enough to move tiny elements):
        # even steps — B's own backward leaves flat views (the optimiser's fast path), overwritten in place with A's values; odd steps — foreign
        # gradient tensors (the gather path)
        mb.set_mask_inputs(forced_mask=(1 - o[2]).float())
        _loss(mb(src, src, None, epoch), src, args, epoch)[0].backward()
        for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            assert (pa.grad is None) == (pb.grad is None), k
            if pa.grad is not None:
                # (pa.grad was clipped in place above: undo nothing — B clips itself, so hand it the UNCLIPPED values)
                if step % 2 == 0:
                    pb.grad.copy_(raw[k])
                    assert ob._flat_grad() is mb._last_gflat, "flat-view gradients must take the optimiser's no-gather path"
                else:
                    pb.grad = raw[k].clone()
        ob.step()
        assert abs(ob.grad_norm() - float(norm)) <= 1e-4 * float(norm)
        if step == 1:
            for gr in oa.param_groups + ob.param_groups:
                gr["lr"] *= 0.3
        err = float((ma.flat - mb.flat).abs().max()) / float(ma.flat.abs().max())
        parity("clip_adam_vs_torch_step%d" % step, err)
        assert err < 2e-6, (step, err)
    assert (ob.tA, ob.tB) == (6, 3)
    sd_o = ob.state_dict()
    oc = ClipAdam(mb.parameters(), lr=1.0, max_grad_norm=0.0)
    oc.load_state_dict(sd_o)
    assert (oc.tA, oc.tB) == (6, 3) and torch.equal(oc.m, ob.m) and oc.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
    never = [k for k, _ in mb.named_parameters() if k.startswith("decoder.time_feature1_.")]
    for k in never:
        assert torch.equal(dict(mb.named_parameters())[k].detach().cpu(), sd[k])
