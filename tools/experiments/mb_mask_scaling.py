"""Mask generation vs the number of cells (data parallel with global masks selects over world x B*T*N cells on every rank)."""
import sys
sys.path.insert(0, ".")
import torch
from gptst_amd import ops, synth
dev = "cuda:0"
def t(f0, n=50):
    for _ in range(3): f0()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        f0()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for W in (1, 2, 4, 8):
    M = W * 32 * 12 * 170
    g = torch.Generator().manual_seed(W)
    noise, na, nr = (torch.rand(M, generator=g).to(dev) for _ in range(3))
    label = torch.randint(0, 10, (M,), generator=g).to(torch.int32).to(dev)
    lc = torch.tensor(synth.class_order(10, 2), dtype=torch.int32, device=dev)
    nums = torch.tensor([M // 8, M // 8], dtype=torch.int32, device=dev)
    tr = t(lambda: ops.mask_random(noise, M // 4))
    ta = t(lambda: ops.mask_adaptive(label, None, lc, nums, na, nr, 1, 1))
    print("world %d  M = %7d   random %.1f us   adaptive %.1f us" % (W, M, tr, ta))
