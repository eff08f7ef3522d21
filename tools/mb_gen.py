#!/usr/bin/env python
"""Micro-benchmark of the parameter-generation job table (engine.gen_all) at the bench shape: rows per block of the forward.
usage (GPU box): python tools/mb_gen.py"""
import sys
sys.path.insert(0, '.')
import torch
from gptst_amd import _C, engine, synth
from gptst_amd.config import make_args
from gptst_amd.model import GPTST_Model, xavier_init_

dev = 'cuda:0'
args = make_args("PEMS08", scaler_zeros=synth.scaler_zeros(), device=dev)
model = xavier_init_(GPTST_Model(args)).to(dev)
p = model.param_views()
B, T, N, C = 32, 12, args.num_nodes, args.hidden_dim
src = synth.make_batch(B, T, N, 1, seed=1).to(dev)
tidx = src[:, :, 0, 1:3].contiguous()


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1000)
    return best


for rows in (8, 16, 24, 32, 48, 64):
    _C.lib().call("gptst_tune", 1, rows)
    print("rows per block %2d: gen_all %.1f us" % (rows, bench(lambda: engine.gen_all(p, tidx, (B, T, N, C)))))
