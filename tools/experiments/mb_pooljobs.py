"""Backward pool-job table at the bench shape (12 big dW: 84 MB): dpool jobs only / demb jobs only / both, replayed from a hipGraph.
Measured (round 3): 24.5 / 33.2 / 48.0 us; a streaming read of 84 MB (torch sum) takes 30.7 us.  Three fused one-pass kernels
(row-chunk x slab with dpool atomics; slab x all rows with per-pair barriers; barrier-free wave-per-tile) took 85 / 44.9 / 77 us."""
import sys
sys.path.insert(0, ".")
import torch
from gptst_amd import ops, _C
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
probs = [(384, 16, 4096, 1)] * 8 + [(170, 16, 4096, 3)] * 4
T = []
for (R, K, cols, ns) in probs:
    T.append(dict(emb=torch.randn(R, K, device=dev), pool=torch.randn(K, cols, device=dev), dW=torch.randn(ns * R, cols, device=dev),
                  dp=torch.zeros(K, cols, device=dev), ns=ns))
de = {384: torch.zeros(384, 16, device=dev), 170: torch.zeros(170, 16, device=dev)}
def run(pool, emb):
    J = ops.PoolJobs()
    for t in T:
        if pool: J.bwd_pool(t["emb"], t["dW"], t["dp"], nsplit=t["ns"])
        if emb: J.bwd_emb(t["dW"], t["pool"], de[t["emb"].shape[0]], nsplit=t["ns"])
    J.launch()
def t(f0, n=100):
    for _ in range(3): f0()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()                      # the host side (24 jobs built in Python) takes longer than the kernels: replay a graph
    with torch.cuda.graph(gr):
        f0()
    f = gr.replay
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
mb = sum(t_["dW"].numel() * 4 for t_ in T) / 1e6
print("dW bytes %.1f MB" % mb)
for name, a in [("dpool only", (1, 0)), ("demb only", (0, 1)), ("both", (1, 1))]:
    us = t(lambda: run(*a))
    print("%-16s %.1f us  (%.2f TB/s of dW)" % (name, us, mb / us))
big = torch.randn(21 * 1000 * 1000, device=dev)
print("torch sum of 84 MB: %.1f us" % t(lambda: big.sum()))
