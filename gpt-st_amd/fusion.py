"""The gate of the downstream front end on HIP (reference model/Model.py:5-18 ``Fusion`` + :106 ``lin_test``; csrc/fusion.hip): forward = ONE launch,
backward = one data-path launch + the library's weight-gradient kernels.  ``fusion_gate(F, flow, fusion, lin_test)`` is a drop-in for
``fusion(F, lin_test(flow))`` on CUDA fp32 tensors with C = 64 (other widths take the torch modules on the GPU; CPU tensors raise); the encoder embedding F is treated as a
constant (the pretrained encoder is frozen, model/Model.py:93-94)."""
import torch

from . import _C, ops
from .ops import MODE_SHARED, _call, _p


class _FusionGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, src, base, Ws, bs, Wh, bh, Wo, bo, Wt, bt):
        rows, C = F.shape
        lda = src.shape[-1]
        need_grad = any(t.requires_grad for t in (Ws, bs, Wh, bh, Wo, bo, Wt, bt))
        out = torch.empty_like(F)
        z = torch.empty_like(F) if need_grad else None
        _call("gptst_fusion_gate_fwd", _p(F), _p(src), lda, base, _p(Ws), _p(bs), _p(Wh), _p(bh), _p(Wo), _p(bo), _p(Wt), _p(bt), _p(out), _p(z),
              rows, C, nbytes=F.numel() * 8)
        ctx.base, ctx.lda = base, lda
        ctx.save_for_backward(F, src, z, Wh, Wo, Wt, bt)
        return out

    @staticmethod
    def backward(ctx, dout):
        F, src, z, Wh, Wo, Wt, bt = ctx.saved_tensors
        rows, C = F.shape
        base = ctx.base
        dout = dout.contiguous()
        dpre, dxd, Hm, xt = (torch.empty_like(F) for _ in range(4))
        _call("gptst_fusion_gate_bwd", _p(dout), _p(F), _p(z), _p(src), ctx.lda, base, _p(Wo), _p(Wt), _p(bt), _p(dpre), _p(dxd), _p(Hm), _p(xt),
              rows, C, nbytes=F.numel() * 28)

        def wb(A, D):
            """nn.Linear gradients of y = A W^T + b from dY = D: (dW [out][in], db) — rows [A^T D | column sums of D] summed over the row splits"""
            part, ns = ops.wgrad(A, D, MODE_SHARED, 1, rows, colsum_d=True)
            s = part.view(ns, C * C + C).sum(0)
            return s[:C * C].view(C, C).t(), s[C * C:]
        dWo, dbo = wb(Hm, dout)
        dWs, dbs = wb(F, dpre)
        dWh, dbh = wb(xt, dpre)
        dx = ops.apply(dpre, Wh.contiguous(), MODE_SHARED, 1, rows) + dxd            # x_t's gradient: through HT_fc and through the blend
        flow = src.view(rows, ctx.lda)[:, :base]
        dWt, dbt = dx.t() @ flow, dx.sum(0)
        return None, None, None, dWs, dbs, dWh, dbh, dWo, dbo, dWt, dbt


def fusion_gate(F, source, fusion, lin_test, base):
    """F (..., C): encoder embedding; source (..., base + 2): the raw batch (its first `base` channels are the flow) -> fused embedding (..., C)."""
    C = F.shape[-1]
    if not (F.is_cuda and source.is_cuda):
        raise RuntimeError("gpt-st_amd: the downstream gate runs on the GPU only (no CPU path exists, as for the encoder in front of it)")
    ok = (F.dtype == torch.float32 and C == 64 and base <= 4 and source.dtype == torch.float32
          and not F.requires_grad and source.shape[:-1] == F.shape[:-1])
    if not ok:                                   # other widths (C = 128): the reference's torch modules, on the GPU as before round 4
        return fusion(F, lin_test(source[..., :base]))
    _C.lib()
    Fc, sc = F.contiguous().view(-1, C), source.contiguous()
    out = _FusionGateFn.apply(Fc, sc.view(-1, sc.shape[-1]), base, fusion.HS_fc.weight, fusion.HS_fc.bias, fusion.HT_fc.weight, fusion.HT_fc.bias,
                              fusion.output_fc.weight, fusion.output_fc.bias, lin_test.weight, lin_test.bias)
    return out.view(F.shape)
