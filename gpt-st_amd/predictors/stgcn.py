"""STGCN predictor (Yu et al., IJCAI 2018) as the reference wires it behind the enhanced embedding: ``-mode eval -model STGCN``
(reference model/STGCN/stgcn.py:127-155, constructed in model/Model.py:52-54).  Restated here with the SAME parameter tree — so a
reference checkpoint loads with ``load_state_dict`` — and the same arithmetic, as one functional forward over three small parameter
holders.  Layout inside: (B, C, T, N); every temporal convolution is 'same'-padded (the reference keeps all 12 steps, stgcn.py:44),
so the output is (B, T, N, dim_out).

    ST block:  GLU temporal conv (kt)  ->  Chebyshev graph conv (ks) + ReLU  ->  ReLU temporal conv (kt)  ->  LayerNorm([N, C])
    output:    GLU temporal conv (outputl_ks)  ->  LayerNorm([N, C])  ->  sigmoid 1x1 conv  ->  1x1 conv to dim_out

This is downstream plumbing (torch ops on the GPU), not the hand-written hot path.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Align(nn.Module):
    """Residual branch between channel counts: 1x1 conv down (stgcn.py:15-16), zero-pad up (:21-22), identity otherwise."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.c_in, self.c_out = c_in, c_out
        if c_in > c_out:
            self.conv1x1 = nn.Conv2d(c_in, c_out, 1)

    def forward(self, x):
        if self.c_in > self.c_out:
            return self.conv1x1(x)
        if self.c_in < self.c_out:
            return F.pad(x, (0, 0, 0, 0, 0, self.c_out - self.c_in))
        return x


class _TConv(nn.Module):
    """Temporal convolution over T with kernel (kt, 1), 'same' padding, residual, and one of three gates (stgcn.py:24-53)."""

    def __init__(self, kt, c_in, c_out, act):
        super().__init__()
        self.act, self.c_out = act, c_out
        self.align = _Align(c_in, c_out)
        self.conv = nn.Conv2d(c_in, 2 * c_out if act == "GLU" else c_out, (kt, 1), 1, padding=((kt - 1) // 2, 0))

    def forward(self, x):
        res, y = self.align(x), self.conv(x)
        if self.act == "GLU":
            p, q = y.split(self.c_out, dim=1)
            return (p + res) * torch.sigmoid(q)
        return torch.sigmoid(y + res) if self.act == "sigmoid" else torch.relu(y + res)


class _SConv(nn.Module):
    """Chebyshev graph convolution: sum_k theta[:, :, k] applied to (L_k x) over the nodes, + bias, residual, ReLU (stgcn.py:55-81)."""

    def __init__(self, ks, c_in, c_out, lk):
        super().__init__()
        self.register_buffer("Lk", lk, persistent=False)                 # (ks, N, N); the reference keeps it as a plain attribute
        self.theta = nn.Parameter(torch.empty(c_in, c_out, ks))
        self.b = nn.Parameter(torch.empty(1, c_out, 1, 1))
        self.align = _Align(c_in, c_out)
        nn.init.kaiming_uniform_(self.theta, a=math.sqrt(5))            # stgcn.py:64-68
        fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.theta)
        nn.init.uniform_(self.b, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def forward(self, x):
        xk = torch.einsum("knm,bctm->bctkn", self.Lk, x)                 # node mixing per Chebyshev order
        y = torch.einsum("cok,bctkn->botn", self.theta, xk) + self.b
        return torch.relu(y + self.align(x))


class _STBlock(nn.Module):
    def __init__(self, ks, kt, n, c, p, lk):
        super().__init__()
        self.tconv1 = _TConv(kt, c[0], c[1], "GLU")
        self.sconv = _SConv(ks, c[1], c[1], lk)
        self.tconv2 = _TConv(kt, c[1], c[2], "relu")
        self.ln = nn.LayerNorm([n, c[2]])
        self.dropout = nn.Dropout(p)

    def forward(self, x):
        y = self.tconv2(self.sconv(self.tconv1(x)))
        y = self.ln(y.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)            # normalise over (N, C) per (b, t)   (stgcn.py:98)
        return self.dropout(y)


class _FC(nn.Module):
    def __init__(self, c, out_dim):
        super().__init__()
        self.conv = nn.Conv2d(c, out_dim, 1)

    def forward(self, x):
        return self.conv(x)


class _Head(nn.Module):
    def __init__(self, c, t, n, out_dim):
        super().__init__()
        self.tconv1 = _TConv(t, c, c, "GLU")
        self.ln = nn.LayerNorm([n, c])
        self.tconv2 = _TConv(1, c, c, "sigmoid")
        self.fc = _FC(c, out_dim)

    def forward(self, x):
        y = self.tconv1(x)
        y = self.ln(y.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return self.fc(self.tconv2(y))


class STGCN(nn.Module):
    """``args_predictor``: Ks, Kt, num_nodes, G (Ks, N, N) Chebyshev polynomials of the scaled Laplacian (gptst_amd.graph), blocks1
    = [c0, c1, c2], drop_prob, outputl_ks — the fields of reference STGCN/args.py:52-88."""

    def __init__(self, args_predictor, device, dim_in, dim_out):
        super().__init__()
        a = args_predictor
        lk = a.G.to(device)
        blocks0 = [dim_in, a.blocks1[1], a.blocks1[0]]                    # stgcn.py:134
        self.st_conv1 = _STBlock(a.Ks, a.Kt, a.num_nodes, blocks0, a.drop_prob, lk)
        self.st_conv2 = _STBlock(a.Ks, a.Kt, a.num_nodes, a.blocks1, a.drop_prob, lk)
        self.output = _Head(a.blocks1[2], a.outputl_ks, a.num_nodes, dim_out)

    def forward(self, x):                                                  # x (B, T, N, dim_in)
        y = self.output(self.st_conv2(self.st_conv1(x.permute(0, 3, 1, 2))))
        return y.permute(0, 2, 3, 1)                                      # (B, T, N, dim_out)
