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
        R = ops.tmix(x, G)                                                               # :157-158
        Wbt, bbt = ops.poolgen(te, wpool, bpool)                                         # :160-161
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
        dbias = torch.zeros(BT, C, device=x.device)
        dR = ops.apply(dout, Wbt, MODE_TIME, BT, N, A2=out, transw=True, pro=PRO_DPRE, colsum=dbias)
        dWbt, ns = ops.wgrad(R, dout, MODE_TIME, BT, N, D2=out, pro=PRO_DPRE)
        dwpool, dbpool = torch.zeros_like(wpool), torch.zeros_like(bpool)
        dte = torch.zeros_like(te)
        ops.poolgen_bwd_pool(te, dWbt, dwpool, dbias, dbpool, nsplit=ns)
        ops.poolgen_bwd_emb(dWbt, wpool, dte, dbias, bpool, nsplit=ns)
        dx = ops.tmix(dR, G, dOut=dout, Y=out)
        dG = ops.tmix_dgraph(dR, x)
        dA = ops.gram_bwd(A, dG)
        dadj, dne = torch.zeros_like(adj), torch.zeros_like(node_emb)
        ops.poolgen_bwd_pool(node_emb.contiguous(), dA, dadj)
        ops.poolgen_bwd_emb(dA, adj, dne)
        return dx, dne, dte.view(B, T, d), dadj, dwpool, dbpool


def hypertem(x, node_emb, time_eb, adj, wpool, bpool):
    return HyperTemFn.apply(x, node_emb, time_eb, adj, wpool, bpool)
