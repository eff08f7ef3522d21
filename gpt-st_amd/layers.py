"""Layer-level compositions of the HIP kernels with hand-written backward (torch.autograd.Function)."""
import torch

from . import ops
from .ops import (EPI_PLAIN, EPI_RES_LRELU, MODE_NODE, MODE_SHARED, MODE_TIME, PRO_DPRE, PRO_NONE)


class HyperTemFn(torch.autograd.Function):
    """hyperTem.forward (reference GPTST.py:154-163):  out = LReLU((G_n X) W_bt + b_bt + X)."""

    @staticmethod
    def forward(ctx, x, node_emb, time_eb, adj, wpool, bpool):
        B, T, N, C = x.shape
        d, Hm = adj.shape[0], adj.shape[1]
        x = x.contiguous()
        te = time_eb.reshape(B * T, d).contiguous()
        A = ops.poolgen(node_emb.contiguous(), adj.reshape(d, Hm * T)).view(N, Hm, T)     # :156
        G = ops.gram_fwd(A)
        Wbt, bbt = ops.poolgen(te, wpool, bpool)                                         # :160-161
        if C == 64:                                                                      # fused kernel (hypertem.hip)
            R, out = ops.hypertem_fwd(x, G, Wbt, bbt)
        else:
            R = ops.tmix(x, G)                                                           # :157-158
            out = ops.apply(R, Wbt, MODE_TIME, B * T, N, bias=bbt, resid=x, epi=EPI_RES_LRELU)   # :162-163
        ctx.save_for_backward(x, R, out, A, G, Wbt, te, node_emb, adj, wpool, bpool)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, R, out, A, G, Wbt, te, node_emb, adj, wpool, bpool = ctx.saved_tensors
        B, T, N, C = x.shape
        d, Hm = adj.shape[0], adj.shape[1]
        BT = B * T
        dout = dout.contiguous()
        dWbt, ns = ops.wgrad(R, dout, MODE_TIME, BT, N, D2=out, pro=PRO_DPRE)
        if C == 64:                                                # fused backward: partial bias / graph gradients, no atomics
            dx, dbias, dG = ops.hypertem_bwd(dout, out, x, G, Wbt)
            nsb, nsG = ops.hypertem_ntiles(N), B
        else:
            dR, dbias, nsb = ops.apply(dout, Wbt, MODE_TIME, BT, N, A2=out, transw=True, pro=PRO_DPRE, colsum=True)
            dx = ops.tmix(dR, G, dOut=dout, Y=out)
            dG, nsG = ops.tmix_dgraph(dR, x), 1
        dwpool, dbpool = torch.zeros_like(wpool), torch.zeros_like(bpool)
        dte = torch.zeros_like(te)
        J = ops.PoolJobs()
        J.bwd_pool(te, dWbt.view(ns * BT, C * C), dwpool.view(d, C * C), nsplit=ns)
        J.bwd_pool(te, dbias, dbpool, nsplit=nsb)
        J.bwd_emb(dWbt.view(ns * BT, C * C), wpool.view(d, C * C), dte, nsplit=ns)
        J.bwd_emb(dbias, bpool, dte, nsplit=nsb)
        J.launch()
        dA = ops.gram_bwd(A, dG, nsplit=nsG)
        dadj, dne = torch.zeros_like(adj), torch.zeros_like(node_emb)
        ops.poolgen_bwd_pool(node_emb.contiguous(), dA, dadj)
        ops.poolgen_bwd_emb(dA, adj, dne)
        return dx, dne, dte.view(B, T, d), dadj, dwpool, dbpool


def hypertem(x, node_emb, time_eb, adj, wpool, bpool):
    return HyperTemFn.apply(x, node_emb, time_eb, adj, wpool, bpool)


CAP_BWD_ONE_LAUNCH = False      # tests: CapFn.backward through gptst_cap_cross_route_lin_bwd (the fused step's routing backward) instead of the per-kernel calls


class CapFn(torch.autograd.Function):
    """cap.forward (reference GPTST.py:100-141).  Returns (out, c (B,T,HS,N) detached, dyn (B,HT,T*HS) detached)."""

    @staticmethod
    def forward(ctx, x, node_emb, time_eb_spg, teb, t_adj, adj, wspa, bspa, lnp_w, lnp_b, tmpl, num_route):
        B, T, N, C = x.shape
        ds, HS = adj.shape[0], adj.shape[1]
        HT = t_adj.shape[1]
        BT = B * T
        x = x.contiguous()
        teb2 = teb.reshape(BT, ds).contiguous()
        tes = time_eb_spg.contiguous()
        ne = node_emb.contiguous()
        dadj = ops.poolgen(teb2, adj.reshape(ds, HS * N))                                  # :104
        c, s = ops.cap_route_fwd(x, lnp_w, lnp_b, dadj, HS, num_route)                     # :102-123
        dyn = ops.poolgen(tes, t_adj.reshape(ds, HT * T * HS)).view(B, HT, T * HS)         # :129
        v, Ht, Rt = ops.cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT)                          # :125-134
        rec = ops.cap_rec_fwd(c, v, N, C)                                                  # :135
        Wn, bn = ops.poolgen(ne, wspa, bspa)                                               # :137-138
        out = ops.apply(rec, Wn, MODE_NODE, BT, N, bias=bn, resid=x.view(-1, C), epi=EPI_RES_LRELU).view(B, T, N, C)   # :139-141
        ctx.save_for_backward(x, out, rec, c, s, v, Ht, Rt, dyn, Wn, teb2, tes, ne, t_adj, adj, wspa, bspa, lnp_w, lnp_b, tmpl)
        ctx.mark_non_differentiable(c, dyn)
        return out, c.view(B, T, HS, N), dyn

    @staticmethod
    def backward(ctx, dout, _dc, _ddyn):
        (x, out, rec, c, s, v, Ht, Rt, dyn, Wn, teb2, tes, ne, t_adj, adj, wspa, bspa, lnp_w, lnp_b, tmpl) = ctx.saved_tensors
        B, T, N, C = x.shape
        ds, HS = adj.shape[0], adj.shape[1]
        HT = t_adj.shape[1]
        BT = B * T
        dev = x.device
        dout = dout.contiguous().view(-1, C)
        out2, x2 = out.view(-1, C), x.view(-1, C)
        # node-conditioned apply
        dbn = torch.zeros(N, C, device=dev)
        drec = ops.apply(dout, Wn, MODE_NODE, BT, N, A2=out2, transw=True, pro=PRO_DPRE, colsum=dbn)
        dWn, ns = ops.wgrad(rec, dout, MODE_NODE, BT, N, D2=out2, pro=PRO_DPRE)
        dwspa, dbspa, dne = torch.zeros_like(wspa), torch.zeros_like(bspa), torch.zeros_like(ne)
        ops.poolgen_bwd_pool(ne, dWn, dwspa, dbn, dbspa, nsplit=ns)
        ops.poolgen_bwd_emb(dWn, wspa, dne, dbn, bspa, nsplit=ns)
        # scatter, cross-time, soft assignment
        dc1, dv = ops.cap_rec_bwd(drec, c, v)
        lin = None
        if CAP_BWD_ONE_LAUNCH and C == 64:
            # cross-time backward + routing backward + the entry Linear's backward and the residual branch as ONE launch (the step's default, r05)
            lin = ops.cap_cross_route_lin_bwd(x, lnp_w, lnp_b, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dout, out2, False, B, T, HS, HT,
                                              flags=torch.zeros(4 * B, device=dev))
        if lin is not None:
            dx, dWp, dbp_part, dlogit, ddyn = lin
            dlnp_w, dbp = dWp.view(-1, C, C).sum(0), dbp_part.sum(0, keepdim=True)      # B*T (+ node halves) partial rows
        else:
            dS, ddyn = ops.cap_cross_bwd(dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT)
            dY, dlogit = ops.cap_route_bwd(x, lnp_w, lnp_b, c, dc1, dS)
            # Linear ln_p: Y = X Wp^T + bp
            dbp = torch.zeros(1, C, device=dev)
            dx = ops.apply(dY, lnp_w, MODE_SHARED, BT, N, resid=dout, resid2=out2, epi=ops.EPI_ADD_DPRE, colsum=dbp)
            dWp, ns2 = ops.wgrad(dY, x2, MODE_SHARED, BT, N)
            dlnp_w = dWp.view(ns2, C, C).sum(0)
        dt_adj, dtes = torch.zeros_like(t_adj), torch.zeros_like(tes)
        ops.poolgen_bwd_pool(tes, ddyn, dt_adj)
        ops.poolgen_bwd_emb(ddyn, t_adj, dtes)
        dadj, dteb = torch.zeros_like(adj), torch.zeros_like(teb2)
        ops.poolgen_bwd_pool(teb2, dlogit, dadj)
        ops.poolgen_bwd_emb(dlogit, adj, dteb)
        return (dx.view(B, T, N, C), dne, dtes, dteb.view(B, T, ds), dt_adj, dadj, dwspa, dbspa, dlnp_w, dbp.view(C), None, None)


def cap(x, node_emb, time_eb_spg, teb, t_adj, adj, wspa, bspa, lnp_w, lnp_b, tmpl, num_route):
    return CapFn.apply(x, node_emb, time_eb_spg, teb, t_adj, adj, wspa, bspa, lnp_w, lnp_b, tmpl, num_route)


class CondLinearFn(torch.autograd.Function):
    """LReLU(x @ W_g + b_g) with embedding-generated weights, no residual (MLP_RL, reference GPTST.py:24-32).
    mode MODE_NODE: g = n, emb (N,d);  mode MODE_TIME: g = (b,t), emb (B*T,d)."""

    @staticmethod
    def forward(ctx, x, emb, wpool, bpool, mode):
        B, T, N, C = x.shape
        x = x.contiguous()
        emb2 = emb.reshape(-1, emb.shape[-1]).contiguous()
        Wg, bg = ops.poolgen(emb2, wpool, bpool)
        out = ops.apply(x.view(-1, C), Wg, mode, B * T, N, bias=bg, epi=ops.EPI_LRELU).view(B, T, N, C)
        ctx.save_for_backward(x, out, Wg, emb2, wpool, bpool)
        ctx.mode, ctx.emb_shape = mode, emb.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, Wg, emb2, wpool, bpool = ctx.saved_tensors
        B, T, N, C = x.shape
        BT, mode = B * T, ctx.mode
        G = emb2.shape[0]
        dout = dout.contiguous().view(-1, C)
        db = torch.zeros(G, C, device=x.device)
        dx = ops.apply(dout, Wg, mode, BT, N, A2=out.view(-1, C), transw=True, pro=PRO_DPRE, colsum=db)
        dW, ns = ops.wgrad(x.view(-1, C), dout, mode, BT, N, D2=out.view(-1, C), pro=PRO_DPRE)
        dwpool, dbpool, demb = torch.zeros_like(wpool), torch.zeros_like(bpool), torch.zeros_like(emb2)
        ops.poolgen_bwd_pool(emb2, dW, dwpool, db, dbpool, nsplit=ns)
        ops.poolgen_bwd_emb(dW, wpool, demb, db, bpool, nsplit=ns)
        return dx.view(B, T, N, C), demb.view(ctx.emb_shape), dwpool, dbpool, None


def cond_linear(x, emb, wpool, bpool, mode):
    return CondLinearFn.apply(x, emb, wpool, bpool, mode)
