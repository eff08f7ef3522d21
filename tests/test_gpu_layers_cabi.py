"""Layer-level C-ABI entry points (csrc/layers.hip, SURVEY.md 8b "minimum set"): ONE call per reference layer, called here through ctypes
with raw device pointers — exactly what a non-Python consumer would do — and compared with the pinned oracle (forward + every gradient)."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_kernels import _cap_case, _hypertem_case, close, rnd  # noqa: E402
from oracle import gptst_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from gptst_amd import _C
    return _C.lib(), _C.stream()


def _regions(kind, B, T, N, C, d, Hm, ds, HS, HT, base):
    lib, _ = _lib()
    sv, sc = ctypes.c_long(0), ctypes.c_long(0)
    lib.call("gptst_layer_bytes", kind, B, T, N, C, d, Hm, ds, HS, HT, base, ctypes.byref(sv), ctypes.byref(sc))
    return (torch.empty(max(sv.value, 4) // 4, device=DEV), sv.value), (torch.empty(max(sc.value, 4) // 4, device=DEV), sc.value)


def _p(t):
    assert t.is_contiguous()
    return t.data_ptr()


@pytest.mark.parametrize("B,N,d,Hm", [(2, 20, 8, 8), (3, 170, 16, 8), (1, 33, 4, 5)])
def test_hypertem_layer_c_entry(B, N, d, Hm):
    lib, st = _lib()
    C, T = 64, 12
    ts, go = _hypertem_case(B, N, C, d, Hm, 11)
    cpu = [t.clone().requires_grad_() for t in ts]
    sd = {"h.adj": cpu[3], "h.weights_pool": cpu[4], "h.bias_pool": cpu[5]}
    ref = O.hypertem(sd, "h.", cpu[0], cpu[1], cpu[2])
    go = go * (ref.detach().abs() > 1e-5)
    (ref * go).sum().backward()
    x, ne, te, adj, wp, bp = [t.to(DEV).contiguous() for t in ts]
    (saved, svb), (scr, scb) = _regions(0, B, T, N, C, d, Hm, 0, 0, 0, 0)
    out = torch.empty_like(x)
    lib.call("gptst_hypertem_layer_fwd", _p(x), _p(ne), _p(te), _p(adj), _p(wp), _p(bp), _p(out), _p(saved), svb, B, T, N, C, d, Hm, st)
    close(out, ref, what="hypertem layer out")
    dx = torch.empty_like(x)
    grads = [torch.zeros_like(t) for t in (ne, te, adj, wp, bp)]
    dout = go.to(DEV).contiguous()
    lib.call("gptst_hypertem_layer_bwd", _p(dout), _p(x), _p(out), _p(ne), _p(te), _p(adj), _p(wp), _p(bp), _p(saved), svb, _p(dx),
             *[_p(g_) for g_ in grads], _p(scr), scb, B, T, N, C, d, Hm, st)
    for nm, a, b in zip(["x", "node_emb", "time_eb", "adj", "wpool", "bpool"], [dx] + grads, cpu):
        close(a, b.grad, what="hypertem layer d" + nm)
    # a region that is too small is refused, nothing is launched
    with pytest.raises(Exception):
        lib.call("gptst_hypertem_layer_fwd", _p(x), _p(ne), _p(te), _p(adj), _p(wp), _p(bp), _p(out), _p(saved), svb - 256, B, T, N, C, d, Hm, st)


@pytest.mark.parametrize("B,N,d,ds,HS,HT,R", [(2, 20, 8, 4, 5, 6, 3), (2, 170, 16, 4, 10, 16, 2), (1, 33, 4, 3, 16, 5, 0)])
def test_cap_layer_c_entry(B, N, d, ds, HS, HT, R):
    lib, st = _lib()
    C, T = 64, 12
    ts, go = _cap_case(B, N, C, d, ds, HS, HT, 21)
    tmpl = torch.linspace(1, T, steps=T) / 12.0
    cpu = [t.clone().requires_grad_() for t in ts]
    sd = {"c.t_adj": cpu[4], "c.adj": cpu[5], "c.weights_spa": cpu[6], "c.bias_spa": cpu[7], "c.ln_p.weight": cpu[8],
          "c.ln_p.bias": cpu[9], "c.mask_template": tmpl}
    ref, cref, dynref, aux = O.cap(sd, "c.", cpu[0], cpu[1], cpu[2], cpu[3], R, materialize_5d=(N <= 64), return_aux=True)
    go = go * (ref.detach().abs() > 1e-5)
    (ref * go).sum().backward()
    x, ne, tes, teb, t_adj, adj, wspa, bspa, lw, lb = [t.to(DEV).contiguous() for t in ts]
    tm = tmpl.to(DEV)
    (saved, svb), (scr, scb) = _regions(1, B, T, N, C, d, 0, ds, HS, HT, 0)
    out, c = torch.empty_like(x), torch.empty(B * T, HS, N, device=DEV)
    dyn = torch.empty(B, HT, T * HS, device=DEV)
    lib.call("gptst_cap_layer_fwd", _p(x), _p(ne), _p(tes), _p(teb), _p(lw), _p(lb), _p(adj), _p(t_adj), _p(wspa), _p(bspa), _p(tm), _p(out),
             _p(c), _p(dyn), _p(saved), svb, B, T, N, C, d, ds, HS, HT, R, st)
    close(c.view(B, T, HS, N), cref.squeeze(-1), what="cap layer c")
    close(dyn, dynref, what="cap layer dyn")
    close(out, ref, what="cap layer out")
    dx = torch.empty_like(x)
    g = {k: torch.zeros_like(v) for k, v in dict(ne=ne, tes=tes, teb=teb, lw=lw, lb=lb, adj=adj, t_adj=t_adj, wspa=wspa, bspa=bspa).items()}
    dout = go.to(DEV).contiguous()
    lib.call("gptst_cap_layer_bwd", _p(dout), _p(x), _p(out), _p(c), _p(dyn), _p(ne), _p(tes), _p(teb), _p(lw), _p(lb), _p(adj), _p(t_adj),
             _p(wspa), _p(bspa), _p(tm), _p(saved), svb, _p(dx), _p(g["ne"]), _p(g["tes"]), _p(g["teb"]), _p(g["lw"]), _p(g["lb"]), _p(g["adj"]),
             _p(g["t_adj"]), _p(g["wspa"]), _p(g["bspa"]), _p(scr), scb, B, T, N, C, d, ds, HS, HT, st)
    got = [dx, g["ne"], g["tes"], g["teb"], g["t_adj"], g["adj"], g["wspa"], g["bspa"], g["lw"], g["lb"]]
    names = ["x", "node_emb", "time_eb_spg", "teb", "t_adj", "adj", "wspa", "bspa", "lnp_w", "lnp_b"]
    for nm, a, b in zip(names, got, cpu):
        close(a, b.grad, what="cap layer d" + nm)


@pytest.mark.parametrize("B,N,d,HS,base", [(2, 20, 8, 6, 1), (2, 170, 16, 10, 1), (1, 33, 4, 5, 2)])
def test_mlprl_layer_c_entry(B, N, d, HS, base):
    lib, st = _lib()
    C, T = 64, 12
    g = torch.Generator().manual_seed(71)
    src = rnd(B, T, N, base + 2, g=g)
    te, ne = rnd(B, T, d, g=g), rnd(N, d, g=g)
    prm = {"m.ln1.weight": rnd(C, base, g=g, scale=0.5), "m.ln1.bias": rnd(C, g=g, scale=0.3),
           "m.weights_pool_spa": rnd(d, C, C, g=g, scale=0.1), "m.bias_pool_spa": rnd(d, C, g=g, scale=0.3),
           "m.weights_pool_tem": rnd(d, C, C, g=g, scale=0.1), "m.bias_pool_tem": rnd(d, C, g=g, scale=0.3),
           "m.ln3.weight": rnd(HS, C, g=g, scale=0.3), "m.ln3.bias": rnd(HS, g=g, scale=0.3)}
    go = rnd(B, T, N, HS, g=g)
    cpu = {k: v.clone().requires_grad_() for k, v in prm.items()}
    tec, nec = te.clone().requires_grad_(), ne.clone().requires_grad_()
    ref = O.mlp_rl(cpu, "m.", src[..., :base], tec, nec)
    (ref * go).sum().backward()
    dp = {k: v.to(DEV).contiguous() for k, v in prm.items()}
    a, ted, ned = src.to(DEV).contiguous(), te.to(DEV).reshape(B * T, d).contiguous(), ne.to(DEV).contiguous()
    (saved, svb), (scr, scb) = _regions(2, B, T, N, C, d, 0, 0, HS, 0, base)
    logits = torch.empty(B * T * N, HS, device=DEV)
    lib.call("gptst_mlprl_layer_fwd", _p(a), base + 2, _p(ted), _p(ned), _p(dp["m.ln1.weight"]), _p(dp["m.ln1.bias"]),
             _p(dp["m.weights_pool_spa"]), _p(dp["m.bias_pool_spa"]), _p(dp["m.weights_pool_tem"]), _p(dp["m.bias_pool_tem"]),
             _p(dp["m.ln3.weight"]), _p(dp["m.ln3.bias"]), _p(logits), _p(saved), svb, B, T, N, C, d, base, HS, st)
    close(logits.view(B, T, N, HS), ref, what="mlprl layer logits")
    gr = {k: torch.zeros_like(v) for k, v in dp.items()}
    dte, dne = torch.zeros_like(ted), torch.zeros_like(ned)
    dl = go.to(DEV).reshape(-1, HS).contiguous()
    lib.call("gptst_mlprl_layer_bwd", _p(dl), _p(a), base + 2, _p(ted), _p(ned), _p(dp["m.ln1.weight"]), _p(dp["m.weights_pool_spa"]),
             _p(dp["m.bias_pool_spa"]), _p(dp["m.weights_pool_tem"]), _p(dp["m.bias_pool_tem"]), _p(dp["m.ln3.weight"]), _p(saved), svb,
             _p(dte), _p(dne), _p(gr["m.ln1.weight"]), _p(gr["m.ln1.bias"]), _p(gr["m.weights_pool_spa"]), _p(gr["m.bias_pool_spa"]),
             _p(gr["m.weights_pool_tem"]), _p(gr["m.bias_pool_tem"]), _p(gr["m.ln3.weight"]), _p(gr["m.ln3.bias"]), _p(scr), scb,
             B, T, N, C, d, base, HS, st)
    close(dte.view(B, T, d), tec.grad, what="mlprl layer dtime_eb")
    close(dne, nec.grad, what="mlprl layer dnode_emb")
    for k in prm:
        close(gr[k], cpu[k].grad, what="mlprl layer d" + k.split(".", 1)[1])
