#!/usr/bin/env python
"""Downstream gate (reference model/Model.py:5-18 Fusion + :106 lin_test): the fused HIP launch(es) against the torch modules at the bench shape
(B, T, N, C) = (32, 12, 170, 64) — and at the reference's eval batch 64.  usage (GPU box, repo root): python tools/mb_fusion.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptst_amd.enhance import Fusion          # noqa: E402
from gptst_amd.fusion import fusion_gate     # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for B in (32, 64):
    F, src, go = torch.randn(B, 12, 170, 64, device=dev), torch.randn(B, 12, 170, 3, device=dev), torch.randn(B, 12, 170, 64, device=dev)
    fus, lin = Fusion(64).to(dev), torch.nn.Linear(1, 64).to(dev)
    params = list(fus.parameters()) + list(lin.parameters())

    def t_fwd():
        with torch.no_grad():
            return fus(F, lin(src[..., :1]))

    def h_fwd():
        with torch.no_grad():
            return fusion_gate(F, src, fus, lin, 1)

    def t_fb():
        for p in params:
            p.grad = None
        (fus(F, lin(src[..., :1])) * go).sum().backward()

    def h_fb():
        for p in params:
            p.grad = None
        (fusion_gate(F, src, fus, lin, 1) * go).sum().backward()
    A = F.numel() * 4 / 1e6
    print("B = %d (A = %.1f MB): forward   torch %.1f us   HIP %.1f us   (one pass over F + out = %.1f us at 8 TB/s)" % (B, A, timeit(t_fwd), timeit(h_fwd), 2 * A / 8))
    print("B = %d: forward + backward (incl. the loss stand-in)   torch %.1f us   HIP %.1f us" % (B, timeit(t_fb), timeit(h_fb)))
