"""Rank-local cost of data parallelism with global masks, on ONE GPU: the step is run as rank 0 of a pretended world of W ranks whose
collectives are replaced by local copies (so RCCL time is NOT included): mask noise / label vector / selections over W x B*T*N cells."""
import sys, time
sys.path.insert(0, ".")
import torch
from gptst_amd import synth
from gptst_amd.config import make_args
from gptst_amd.model import GPTST_Model, xavier_init_
from gptst_amd.step import PretrainStep

class FakeDP:
    capturable = True
    def __init__(self, world): self.world, self.rank = world, 0
    def allreduce_(self, buf): return buf
    def gather_labels(self, local, out=None):
        out.view(self.world, -1).copy_(local.view(1, -1).expand(self.world, -1))      # ONE launch, as the real all-gather is (W copies cost 45 us at W = 8)
        return out
    def rows_of(self, flat_global, per_rank): return flat_global[:per_rank]
    def barrier(self): pass

dev = "cuda:0"
for W in (1, 2, 4, 8):
    args = make_args("PEMS08", scaler_zeros=synth.scaler_zeros(), device=dev)
    model = xavier_init_(GPTST_Model(args)).to(dev)
    st = PretrainStep(model, args, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=32, use_graph=True, dp=FakeDP(W) if W > 1 else None, seed=7)
    src = synth.make_batch(32, 12, args.num_nodes, args.input_base_dim, interval=args.interval, seed=2024).to(dev)
    st.src.copy_(src)
    res = []
    for epoch in (200, 1):
        srcs = st.group_sources(4)
        for s_ in srcs: s_.copy_(src)
        for _ in range(6): st.step_group(srcs, epoch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): st.step_group(srcs, epoch)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 200 * 1e6)
    print("world %d: adaptive %.1f us/step, random %.1f us/step" % (W, res[0], res[1]))
