"""Layer chain only (encoder + decoder forward, loss head, backward chain; no generation / masks / reductions / optimiser):
K branches of B/K samples inside ONE hipGraph vs one branch of B.   usage: python scratch/exp_split2.py"""
import sys
import time
sys.path.insert(0, '.')
import torch
from gptst_amd import synth, ops, engine
from gptst_amd.config import make_args
from gptst_amd.model import GPTST_Model, init_seed, xavier_init_

dev = torch.device('cuda:0')
args = make_args('PEMS08', scaler_zeros=synth.scaler_zeros(), device=str(dev))
init_seed(args.seed)
T, N, C, base = 12, args.num_nodes, args.hidden_dim, args.input_base_dim
model = xavier_init_(GPTST_Model(args)).to(dev)
p = model.param_views()
gflat = torch.zeros(model.flat.numel(), device=dev)
g = model.views_of(gflat)


def chain(src, gen, tidx, dims, keep):
    M = dims[0] * T * N
    mask = (torch.rand(M * base, device=dev) > 0.25).float()
    red = engine.Reductions()
    emb, c1, tidx, sv_e = engine.model_fwd(p, src, mask, dims, base, model.num_route, model.scaler_zeros, gen=gen[engine.ENC], tidx=tidx)
    _, dec, sv_d = engine.decoder_fwd(p, tidx, emb, dims, model.num_route, gen=gen[engine.DEC], head=False)
    sws = torch.zeros(ops.tail_parts(M), 4, device=dev)
    out, dd = engine.loss_tail(p, g, dec, src, mask, base, synth.SCALER_STD, synth.SCALER_MEAN, args.mape_thresh, sws, red)
    engine.model_bwd(p, g, src, mask, tidx, sv_e, sv_d, dec, None, None, dims, base, model.scaler_zeros, red, dd=dd)
    keep.append((red, sv_e, sv_d, out, dd, mask, sws, emb, dec))


def run(K, B, steps=200):
    Bk = B // K
    dims = (Bk, T, N, C)
    srcs = [synth.make_batch(Bk, T, N, base, interval=args.interval, seed=2024 + i).to(dev) for i in range(K)]
    tidxs = [s[:, :, 0, base:base + 2].contiguous() for s in srcs]
    gens = [engine.gen_all(p, t, dims) for t in tidxs]
    streams = [torch.cuda.Stream() for _ in range(K)]
    keep = []
    for i in range(K):
        chain(srcs[i], gens[i], tidxs[i], dims, keep)
    torch.cuda.synchronize()
    keep.clear()
    gr = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            if K == 1:
                chain(srcs[0], gens[0], tidxs[0], dims, keep)
            else:
                for i, s in enumerate(streams):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        chain(srcs[i], gens[i], tidxs[i], dims, keep)
                for s in streams:
                    cur.wait_stream(s)
    torch.cuda.synchronize()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gr.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("K=%d x B=%d: layer chain %.1f us per %d samples" % (K, Bk, el / steps * 1e6, B), flush=True)


for K in (1, 2, 4):
    run(K, 32)
run(1, 16)
run(1, 8)
run(2, 64)
run(1, 64)
