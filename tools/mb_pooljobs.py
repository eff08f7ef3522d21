#!/usr/bin/env python
"""Where the reduction-job launches of a step spend their time (pool_jobs [3115] 76.8 us at the tail of the step).
One eager step at the bench shape records every PoolJobs.launch() table; each table is then replayed back to back as a whole, split by job
kind, and (for the largest table) job by job.  usage (GPU box): python tools/mb_pooljobs.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                    # noqa: E402
from gptst_amd import ops, synth                                # noqa: E402
from gptst_amd.config import make_args                          # noqa: E402
from gptst_amd.model import GPTST_Model, init_seed, xavier_init_   # noqa: E402
from gptst_amd.step import PretrainStep                         # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
args = make_args("PEMS08", scaler_zeros=synth.scaler_zeros(), device=str(dev))
init_seed(args.seed)
B, T, N = 32, 12, args.num_nodes
model = xavier_init_(GPTST_Model(args)).to(dev)
st = PretrainStep(model, args, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=False, seed=7)
st.src.copy_(synth.make_batch(B, T, N, args.input_base_dim, interval=args.interval, seed=2024).to(dev))
st.step(st.src, 200)
torch.cuda.synchronize()

TABLES = []
orig = ops.PoolJobs.launch


def rec_launch(self):
    if self.jobs:
        TABLES.append(list(self.jobs))
    return orig(self)


ops.PoolJobs.launch = rec_launch
st.step(st.src, 200)
torch.cuda.synchronize()
ops.PoolJobs.launch = orig
KIND = {0: "FWD", 1: "BWD_POOL", 2: "BWD_EMB", 3: "GRAM"}


def run(jobs, reps=REPS):
    """reps launches of the table captured in ONE hipGraph (the ctypes marshalling of a 100-job table costs more host time than the kernel runs)"""
    pj = ops.PoolJobs()
    pj.jobs = list(jobs); pj.launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            pj.jobs = list(jobs); pj.launch()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


def nbytes(j):
    kind, emb, x, pool, out, R, K, cols, ns, ldx = j
    if kind == 0:
        return 4 * R * cols
    if kind == 3:
        return 4 * R * 144
    return 4 * R * ns * cols


for ti, tb in enumerate(TABLES):
    kinds = sorted(set(j[0] for j in tb))
    tot = sum(nbytes(j) for j in tb)
    t_all = run(tb)
    print("table %d: %3d jobs  kinds %s  %.1f MB streamed (each dW counted per job)  all: %.1f us  (%.2f TB/s)" % (
        ti, len(tb), [KIND[k] for k in kinds], tot / 1e6, t_all, tot / t_all / 1e6))
    for k in kinds:
        sub = [j for j in tb if j[0] == k]
        t = run(sub)
        nb = sum(nbytes(j) for j in sub)
        print("    only %-8s %3d jobs %.1f MB: %.1f us (%.2f TB/s)" % (KIND[k], len(sub), nb / 1e6, t, nb / t / 1e6))
if os.environ.get("MB_SKIP_JOBS") == "1":
    sys.exit(0)
big = max(TABLES, key=lambda tb: sum(nbytes(j) for j in tb))
print("largest table, job by job (stand-alone launch each):")
rows = []
for j in big:
    kind, emb, x, pool, out, R, K, cols, ns, ldx = j
    rows.append((run([j], reps=10), KIND[kind], R, K, cols, ns, ldx, nbytes(j) / 1e6))
for t, k, R, K, cols, ns, ldx, mb in sorted(rows, reverse=True)[:40]:
    print("    %-8s R=%4d K=%2d cols=%5d ns=%d ldx=%5d  %.2f MB  %.1f us" % (k, R, K, cols, ns, ldx, mb, t))
print("    ... %d jobs, sum of stand-alone times %.1f us" % (len(rows), sum(r[0] for r in rows)))
