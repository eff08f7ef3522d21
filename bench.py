#!/usr/bin/env python
"""Headline benchmark: GPT-ST pretraining steps/s on PEMS08-shaped synthetic input (B=32, T=12, N=170, C=64) on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = zero_grad -> forward (mask generation included) -> masked-MAE + 0.1 KL -> backward -> clip -> Adam
(reference BasicTrainer.py:72-103) at epoch 200 of 300 (adaptive masking + KL: the phase 290 of the 300 epochs run in).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, measured with
HIP events inside this run) and `cpu_baseline` (the oracle's op-for-op PyTorch-CPU restatement on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The bench prints exactly ONE line on stdout.  RCCL writes its version banner (and at WARN its warnings) to STDOUT when NCCL_DEBUG is exported
# (NCCL_DEBUG=VERSION on the GPU boxes; NCCL_DEBUG_FILE does not move the banner), and through C stdio it lands AFTER the JSON line when
# stdout is a pipe: the variable is dropped for this process and its ranks unless GPTST_KEEP_NCCL_DEBUG=1.
if os.environ.get("GPTST_KEEP_NCCL_DEBUG", "0") != "1":
    os.environ.pop("NCCL_DEBUG", None)

import torch  # noqa: E402

HBM_PEAK = 8.0e12          # B/s   (MI355X_MICROARCH.md: 8 TB/s spec)
MFMA_F32_PEAK = 157.3e12   # FLOP/s fp32 matrix (same file)


def kernel_flops(name, tag, d):
    """Algorithmic FLOPs per launch of the heavy kernels (SURVEY.md §8d formulas)."""
    B, T, N, C, HS, R = d["B"], d["T"], d["N"], d["C"], d["HS"], d["R"]
    rows = B * T * N
    if name in ("gptst_apply", "gptst_wgrad"):
        return 2.0 * rows * C * C
    if name == "gptst_hypertem_fwd":
        return 2.0 * rows * C * C + 2.0 * rows * C * T
    if name == "gptst_hypertem_bwd":                      # dR = dPre W^T, dX = G^T dR, dG = dR X^T
        return 2.0 * rows * C * C + 4.0 * rows * C * T
    if name == "gptst_hypertem_bwd_wgrad":                # + dW_bt = R^T dPre in the same launch
        return 4.0 * rows * C * C + 4.0 * rows * C * T
    if name == "gptst_hypertem_bwd_pair":                 # two layers' backward + weight gradients in one launch (r04)
        return 2.0 * (4.0 * rows * C * C + 4.0 * rows * C * T)
    if name == "gptst_hypertem_chain_fwd":                # tag "x<layers>": consecutive hyperTem layers
        nl = int(tag.split()[-1][1:])
        return nl * (2.0 * rows * C * C + 2.0 * rows * C * T)
    if name in ("gptst_tmix", "gptst_tmix_dgraph"):
        return 2.0 * rows * C * T
    if name == "gptst_cap_route_fwd":
        return 2.0 * rows * C * C + (2 * R + 2) * 2.0 * B * T * HS * N * C
    if name in ("gptst_cap_route_bwd", "gptst_cap_cross_route_bwd"):
        return 2.0 * rows * C * C + 2 * 2.0 * B * T * HS * N * C
    if name in ("gptst_cap_cross_route_lin_bwd", "gptst_cap_cross_route_lin_bwd_jobs", "gptst_cap_cross_route_lin_bwd_split"):   # (_jobs: + carried reduction jobs — bandwidth work, no FLOPs counted)
        # r05: + the entry Linear's backward: dX = dY Wp and dWp = dY^T X on top of the routing backward
        return 6.0 * rows * C * C + 2 * 2.0 * B * T * HS * N * C
    if name in ("gptst_apply_wgrad", "gptst_linear_bwd"):
        return 4.0 * rows * C * C
    if name in ("gptst_cap_rec_fwd",):
        return 2.0 * B * T * HS * N * C
    if name in ("gptst_cap_rec_bwd",):
        return 4.0 * B * T * HS * N * C
    return 0.0


# C-ABI entry point (+ tag) -> kernel symbol prefix in rocprofv3 traces / profiles/pmc_traffic.json
KERNEL_SYMBOL = {
    "gptst_cap_route_fwd": "void cap_route_fwd", "gptst_cap_route_bwd": "void cap_route_bwd2_kernel<64",
    "gptst_hypertem_fwd": "hypertem_fwd_kernel", "gptst_hypertem_bwd": "hypertem_bwd_kernel", "gptst_wgrad": "void wgrad64_kernel",
    "gptst_hypertem_bwd_wgrad": "void hypertem_bwd_wgrad_kernel<", "gptst_cap_cross_route_bwd": "void cap_route_bwd2_kernel<64",
    "gptst_cap_cross_route_lin_bwd": "void cap_route_bwd2_kernel<64", "gptst_cap_cross_route_lin_bwd_jobs": "void cap_route_bwd2_kernel<64", "gptst_cap_cross_route_lin_bwd_split": "void cap_route_bwd2_kernel<64", "gptst_hypertem_bwd_pair": "void hypertem_bwd_pair_kernel<", "gptst_hypertem_chain_fwd": "void hypertem_chain_fwd_kernel<",
    "gptst_cap_cross_rec_fwd": "void cap_cross_rec_fwd_kernel<64>", "gptst_apply_wgrad": "void applywg64_kernel<0,", "gptst_linear_bwd": "void applywg64_kernel<1,",
    "gptst_apply": "void apply64_kernel<", "gptst_tmix": "void tmix_kernel<64", "gptst_tmix_dgraph": "void tmix_dgraph_kernel<64>",
    "gptst_cap_rec_bwd": "void cap_rec_bwd2_kernel<64>", "gptst_cap_cross_bwd": "void cap_cross_bwd_kernel<64>",
    "gptst_cap_rec_fwd": "void cap_rec_fwd_kernel<64>",
}


def kernel_source_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.h + the C-ABI header): identifies the code a PMC pass was collected on.  (The snapshot
    on the GPU box has no .git, so a commit id is not available there; tools/pmc_to_json.py stores this hash next to the counters.)"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "gpt-st_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "gptst_hip.h"), "rb").read())
    return h.hexdigest()[:16]


# SURVEY.md 8(d) algorithmic bytes per launch in units of A = 4*B*T*N*C: a big layer's forward reads X and writes Y (2A), its backward reads
# dY and the saved X and writes dX (3A).  hyperTem is one launch per direction; a cap layer's 2A / 3A are spread over its launches: X read
# by the routing kernels, the layer output written by the node-conditioned apply, dOut read by its backward, dX written by the entry-Linear backward.
ALG_8D_A = {"gptst_hypertem_fwd": 2.0, "gptst_hypertem_bwd": 3.0, "gptst_hypertem_bwd_wgrad": 3.0, "gptst_hypertem_bwd_pair": 6.0, "gptst_cap_route_fwd": 1.0,
            "gptst_cap_cross_route_bwd": 1.0, "gptst_cap_cross_route_lin_bwd": 2.0, "gptst_cap_cross_route_lin_bwd_jobs": 2.0, "gptst_cap_cross_route_lin_bwd_split": 2.0, "gptst_cap_route_bwd": 1.0, "gptst_apply": 1.0, "gptst_apply_wgrad": 1.0, "gptst_linear_bwd": 1.0,
            "gptst_cap_cross_rec_fwd": 0.0, "gptst_cap_rec_fwd": 0.0, "gptst_cap_rec_bwd": 0.0}


def graph_avg_us(name, tag, grid_hint=None):
    """Average duration (us) of this kernel under hipGraph replay from the committed rocprofv3 kernel trace (profiles/graph_kernel_us.json, written by
    tools/trace_to_json.py) — None unless the trace was collected on the kernel sources of this run (hash).  Several grids of one symbol (the routing
    backward carries a different job table per launch) are averaged by their call counts."""
    path = os.path.join(ROOT, "profiles", "graph_kernel_us.json")
    sym = KERNEL_SYMBOL.get(name)
    if sym is None or not os.path.exists(path):
        return None
    js = json.load(open(path))
    if js.get("kernel_src_sha") != kernel_source_hash():
        return None
    if name == "gptst_apply":
        sym = "void apply64_kernel<%s, %s>" % (tag.split()[1][3:], tag.split()[2][3:])
    if name == "gptst_wgrad":
        sym = "void wgrad64_kernel<%s," % tag.split()[1][3:4]
    cands = [(k, v) for k, v in js["kernels"].items() if k.startswith(sym)]
    if grid_hint is not None:
        cands = [c for c in cands if grid_hint in c[0]] or cands
    n = sum(v["calls"] for _, v in cands)
    return sum(v["avg_us"] * v["calls"] for _, v in cands) / n if n else None


def pmc_traffic(name, tag, grid_hint=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/pmc_traffic.json) — or None when there is no pass
    for this kernel or the pass was collected on different kernel sources (its kernel_src_sha != kernel_source_hash())."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    sym = KERNEL_SYMBOL.get(name)
    if sym is None or not os.path.exists(path):
        return None, None
    js = json.load(open(path))
    if js.get("kernel_src_sha") != kernel_source_hash():
        return None, "stale: profiles/pmc_traffic.json was collected on kernel sources %s, this run has %s" % (js.get("kernel_src_sha"), kernel_source_hash())
    ks = js["kernels"]
    if name == "gptst_apply":
        m = {"mode0": None, "mode1": None, "mode2": None}
        pro, epi = tag.split()[1][3:], tag.split()[2][3:]
        sym = "void apply64_kernel<%s, %s>" % (pro, epi)
    if name == "gptst_wgrad":
        sym = "void wgrad64_kernel<%s," % tag.split()[1][3:4]
    cands = [(k, v) for k, v in ks.items() if k.startswith(sym)]
    if grid_hint is not None:
        cands = [c for c in cands if grid_hint in c[0]] or cands
    if not cands:
        return None, None
    k, v = max(cands, key=lambda kv: kv[1]["hbm_bytes"])
    return v["hbm_bytes"], k


def time_kernels(stepper, epoch, nsteps=3):
    """Eager (non-graph) steps with a HIP event pair around every kernel launch on the launch stream."""
    from gptst_amd import ops
    stepper.use_graph = False
    stepper.step(stepper.src, epoch)                # warm
    torch.cuda.synchronize()
    ops.TIMER = []
    for _ in range(nsteps):
        stepper.step(stepper.src, epoch)
    torch.cuda.synchronize()
    rec, ops.TIMER = ops.TIMER, None
    agg = {}
    for name, tag, e0, e1, nb in rec:
        k = (name, tag)
        a = agg.setdefault(k, [0.0, 0, nb])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[1] += 1
    return {k: dict(total_s=v[0] / nsteps, launches=v[1] // nsteps, avg_s=v[0] / v[1], bytes=v[2]) for k, v in agg.items()}


def _handoff_timeouts():
    import ctypes
    from gptst_amd import _C
    n = ctypes.c_int(0)
    _C.lib().call("gptst_handoff_timeouts", ctypes.byref(n))
    return n.value


def retime_kernel(name, tag, reps=50):
    """Average duration of ONE launch of (name, tag): its recorded call is enqueued `reps` times back to back on the launch
    stream, bracketed by one HIP event pair (an event pair around a single launch also counts the ~3-5 us dispatch gap, which
    matters for a 20-30 us kernel; back to back the queue stays full because a ctypes enqueue is faster than the kernel).
    The operand buffers of the recorded call are still mapped: nothing releases the caching allocator in between."""
    from gptst_amd import ops, _C
    args = ops.LAST_CALL.get((name, tag))
    if args is None:
        return None
    lib = _C.lib()
    st = _C.stream()
    for _ in range(5):
        lib.call(name, *args, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.call(name, *args, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def module_path(args, B, epoch, dev, steps=40, warmup=6, clip_adam=False):
    """The OTHER way into the HIP kernels, the one INTEGRATION.md section 1 advertises: the reference's own training loop
    (model/BasicTrainer.py:72-103) around the drop-in nn.Module — GPTST_Model.forward (one autograd node over the HIP forward / backward),
    the reference's torch loss closures (Run.py:91-101, lib/metrics.py:11-18, KLDivLoss(sum)), loss.backward(), clip_grad_norm_(5),
    torch.optim.Adam.step(), loss.item().  No fused step, no hipGraph, per-tensor optimiser: what a maintainer gets from the one-line
    import swap alone (r06: the node's forward and backward replay hipGraphs, gptst_amd/module_graph.py).
    clip_adam: the SECOND line a maintainer may change — Run.py:134's torch.optim.Adam -> gptst_amd.optim.ClipAdam(..., max_grad_norm=5), which does the
    loop's clip_grad_norm_ + Adam.step() as the two launches of gptst_clip_adam (the clip_grad_norm_ line goes).  -> dict(steps_per_s, ms_per_step, ...)."""
    from gptst_amd import synth
    from gptst_amd.model import GPTST_Model, xavier_init_
    model = xavier_init_(GPTST_Model(args)).to(dev)
    if clip_adam:
        from gptst_amd.optim import ClipAdam
        opt = ClipAdam(model.parameters(), lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False, max_grad_norm=args.max_grad_norm)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False)         # Run.py:134
    kl = torch.nn.KLDivLoss(reduction="sum")                                                                            # Run.py:132
    src = synth.make_batch(B, 12, args.num_nodes, args.input_base_dim, interval=args.interval, seed=2024).to(dev)
    mean, std, base = synth.SCALER_MEAN, synth.SCALER_STD, args.input_base_dim

    def mae(pred, true, mask_value):                                                                                    # lib/metrics.py:11-18
        m = torch.gt(true, mask_value)
        return torch.mean(torch.abs(torch.masked_select(true, m) - torch.masked_select(pred, m)))

    def one():
        opt.zero_grad()                                                                                                 # BasicTrainer.py:79
        out, _, mask, prob, eb = model(src, src, None, epoch)                                                           # :82
        label = src[..., :base]
        loss = mae((out * std + mean) * mask, (label * std + mean) * mask, args.mape_thresh)                            # Run.py:92-100
        if epoch > args.change_epoch:
            loss = loss + 0.1 * kl(prob.log(), eb)                                                                      # BasicTrainer.py:84-86
        loss.backward()                                                                                                 # :92
        if not clip_adam:
            torch.nn.utils.clip_grad_norm_(model.parameters(), args.max_grad_norm)                                      # :95-96
        opt.step()                                                                                                      # :97
        return loss.item()                                                                                              # :98 (host sync every step)
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = one()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    from gptst_amd import module_graph
    return dict(steps_per_s=steps / el, ms_per_step=1e3 * el / steps, steps=steps, last_loss=last, graphs=bool(module_graph.ENABLED),
                what="reference-style loop (BasicTrainer.py:72-103) over the drop-in GPTST_Model: autograd node on the HIP kernels (forward / backward "
                     "as hipGraph replays) + torch loss + " + ("gptst_amd.optim.ClipAdam (clip + Adam fused)" if clip_adam else "clip_grad_norm_ + torch.optim.Adam")
                     + ", loss.item() every step")


def cpu_baseline(args, B, budget_s=12.0):
    """The oracle (op-for-op CPU restatement of the reference step) timed on the host cores — reported, never the target."""
    from gptst_amd import synth
    from oracle import gptst_oracle as O
    ncpu = os.cpu_count() or 1
    sd = O.init_state_dict(args, 12)
    st = O.Stepper(sd, args, synth.SCALER_MEAN, synth.SCALER_STD)
    src = synth.make_batch(B, 12, args.num_nodes, args.input_base_dim, seed=2024)
    M = B * 12 * args.num_nodes
    inj = dict(noise_a=synth.make_noise(M, 7), noise_r=synth.make_noise(M, 8), list_c=synth.class_order(args.HS, 7))
    # torch's CPU ops oversubscribe badly with hundreds of threads on these small tensors (256 threads: 159 s/step on the
    # GPU box): pick the fastest of a few thread counts with one step each, then time the bounded sample with it.
    best, cores = None, 1
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        st.step(src, 200, **inj)
        t0 = time.perf_counter(); st.step(src, 200, **inj); dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, th
        if dt > 4 * best:
            break
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        st.step(src, 200, **inj)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 40:
            break
    return dict(value=n / el, unit="steps/s", cores=cores, host_cpus=ncpu, kind="port",
                sample="%d full steps (B=%d, epoch 200: adaptive mask + KL) of the same workload in %.1f s, torch %s CPU, %d threads"
                       % (n, B, el, torch.__version__, cores))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec this script under torch.distributed.run (one rank per GPU, RCCL
    over xGMI), rendezvous on 127.0.0.1 and a free port; rank 0's single JSON line is the only thing on stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--exact-warmup", action="store_true",
                    help="run exactly --warmup untimed steps (default: at least ~0.3 s of them, so that the GPU has reached its steady clocks: "
                         "20 timed steps behind 5 warm-up steps measure 815 steps/s, the same 20 steps behind 200 warm-up steps 833)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dataset", default="PEMS08")
    ap.add_argument("--epoch", type=int, default=200, help="epoch whose masking schedule is benchmarked (of 300)")
    ap.add_argument("--nodes", type=int, default=0, help="override num_nodes (BASELINE configs[4]: 4096)")
    ap.add_argument("--hidden", type=int, default=0, help="override hidden_dim (BASELINE configs[4]: 128)")
    ap.add_argument("--hs", type=int, default=0, help="override the cluster count HS (BASELINE configs[3]: sweep 2/5/10/20/40)")
    ap.add_argument("--shard", choices=["batch", "nodes"], default="batch",
                    help="multi-GPU partitioning: batch = data parallel (default, weak scaling); nodes = node sharding of ONE global "
                         "batch (SURVEY §8e row 2 / BASELINE configs[4], strong scaling; eager, no hipGraph)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="data parallel only: weak = every rank steps its own batch of --batch (default); strong = ONE global batch of "
                         "--global-batch samples split evenly over the ranks (SURVEY 8d c3: 256 = 32 x 8)")
    ap.add_argument("--global-batch", type=int, default=256, help="global batch of --scaling strong")
    ap.add_argument("--native-comm", dest="native_comm", action="store_true", default=None,
                    help="per-step collectives on the C-ABI communicator (RCCL enqueues on the launch stream): the whole step incl. its "
                         "collectives and the optimiser is ONE hipGraph per phase, also with several ranks.  DEFAULT for --gpus N > 1 "
                         "(data parallel and node-sharded); --torch-comm keeps torch.distributed collectives between graph replays")
    ap.add_argument("--torch-comm", dest="native_comm", action="store_false")
    ap.add_argument("--repeats", type=int, default=1,
                    help="SURVEY 8(d) protocol: after the contract's timed region (which `value` reports), time the same K steps this many times "
                         "in total and report every rate and their median as `repeat_values` / `median_value` (e.g. --warmup 50 --steps 500 --repeats 5)")
    ap.add_argument("--group", type=int, default=None,
                    help="optimisation steps per hipGraph replay (PretrainStep.step_group): K consecutive steps, each a full step on its own "
                         "input buffer with its own optimiser update, behind ONE host-scalar copy and ONE graph launch (default: the trainer's "
                         "own setting, config.STEPS_PER_REPLAY = 4; 1 = one replay per step)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--path", choices=["step", "module"], default="step",
                    help="step: the fused PretrainStep (headline); module: ONLY the reference-style loop over the drop-in nn.Module (its rate becomes `value`)")
    ap.add_argument("--no-module-path", action="store_true", help="skip the short module-path measurement reported beside the headline")
    a = ap.parse_args()

    from gptst_amd import synth
    from gptst_amd.config import make_args
    from gptst_amd.model import GPTST_Model, init_seed, xavier_init_
    from gptst_amd.step import PretrainStep

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))              # plain `python bench.py --gpus N`: spawn the N ranks ourselves
    if a.scaling == "strong" and a.shard == "batch":
        assert a.global_batch % a.gpus == 0, "--global-batch must divide over the ranks"
        a.batch = a.global_batch // a.gpus
    local = local % max(torch.cuda.device_count(), 1)             # (tests: several ranks may share one GPU over gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dp = None
    force_dp = os.environ.get("GPTST_FORCE_DP") == "1"            # FORCE_DP: exercise the RCCL path with one rank
    if a.native_comm is None:                                      # default: the capturable communicator whenever there is communication,
        a.native_comm = (world > 1 or force_dp) and os.environ.get("GPTST_DIST_BACKEND", "nccl") == "nccl"   # unless ranks share a GPU (gloo tests)
    if world > 1 or force_dp:
        from gptst_amd.dist import DataParallel
        dp = DataParallel(os.environ.get("GPTST_DIST_BACKEND", "nccl"),   # nccl = RCCL over xGMI; gloo only to exercise the path on one GPU
                          native=bool(a.native_comm) and a.shard == "batch")

    if dp is not None and a.shard == "batch" and a.native_comm and dp.native is None:
        a.native_comm = False                                      # the C-ABI communicator could not be set up on some rank (dist.DataParallel): torch.distributed
    over = {}
    if a.nodes:
        over["num_nodes"] = a.nodes
    if a.hidden:
        over["hidden_dim"] = a.hidden
    if a.hs:
        over["HS"] = a.hs
    args = make_args(a.dataset, scaler_zeros=synth.scaler_zeros(), device=str(dev), **over)
    init_seed(args.seed)
    B, T, N, C = a.batch, 12, args.num_nodes, args.hidden_dim
    if a.shard == "nodes":
        # every rank owns N/world nodes of the SAME global batch; parameters are initialised globally and sliced (shard.py)
        from gptst_amd.shard import DistNodeGroup, ShardedPretrainStep, shard_state_dict
        assert N % world == 0, "equal node shards"
        Nl = N // world
        gmodel = xavier_init_(GPTST_Model(args))                                     # same seed on every rank -> same global init
        sd_l = shard_state_dict(gmodel.state_dict(), rank * Nl, (rank + 1) * Nl)
        largs = make_args(a.dataset, scaler_zeros=synth.scaler_zeros(), device=str(dev), **dict(over, num_nodes=Nl))
        model = GPTST_Model(largs)
        model.load_state_dict(sd_l)
        model = model.to(dev)
        del gmodel
        group = DistNodeGroup(rank, world)
        if a.native_comm:
            from gptst_amd.dist import NativeComm
            from gptst_amd.shard import NativeNodeGroup
            group = NativeNodeGroup(NativeComm(rank=rank, world=world))
        stepper = ShardedPretrainStep(model, largs, N, group, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, seed=7,
                                      use_graph=False if a.no_graph else None)     # None: hipGraph when the collectives are capturable (world = 1)
        gsrc = synth.make_batch(B, T, N, args.input_base_dim, interval=args.interval, seed=2024)
        src = gsrc[:, :, rank * Nl:(rank + 1) * Nl].contiguous().to(dev)
    else:
        model = xavier_init_(GPTST_Model(args)).to(dev)
        stepper = PretrainStep(model, args, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=not a.no_graph, dp=dp,
                               seed=7)          # same seed on every rank: global mask noise and class order must agree (dist.py)
        src = synth.make_batch(B, T, N, args.input_base_dim, interval=args.interval, seed=2024 + rank, start_slot=1000 * rank).to(dev)
    stepper.src.copy_(src)                           # inputs resident in HBM before the timed region
    G = max(int(a.group if a.group is not None else getattr(args, "steps_per_replay", 1)), 1)
    if a.shard == "nodes" or a.no_graph or not stepper.group_ok(a.epoch):
        G = 1
    if G > 1:
        gsrcs = stepper.group_sources(G)             # G consecutive steps per graph replay (PretrainStep.step_group): G input buffers,
        for j_, t_ in enumerate(gsrcs):              # holding G DIFFERENT synthetic batches (r03 review: they were G copies of one)
            t_.copy_(src if j_ == 0 else synth.make_batch(B, T, N, args.input_base_dim, interval=args.interval, seed=2024 + rank + 7919 * j_,
                                                          start_slot=1000 * rank + 37 * j_).to(dev))

    def run(n, epoch):
        """enqueue exactly n optimisation steps"""
        for _ in range(n // G if G > 1 else 0):
            stepper.step_group(gsrcs, epoch)
        for _ in range(n % G if G > 1 else n):
            stepper.step(stepper.src, epoch)

    run(max(a.warmup, 1), a.epoch)
    if G > 1 and a.warmup < G:
        run(G, a.epoch)                              # the group graph is captured outside the timed region
    warm_run = max(a.warmup, 1) + (G if (G > 1 and a.warmup < G) else 0)

    def timed(n):
        """n steps bracketed by barrier + synchronize on both sides -> seconds (max over the ranks)"""
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        run(n, a.epoch)
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize()
        e_ = time.perf_counter() - t_
        return dp.max_over_ranks(e_) if dp is not None else e_

    # the contract command's LITERAL number: exactly --steps steps right behind exactly --warmup warm-up steps (a short run times the clock
    # ramp: VERDICT r04) — reported as value_exact_warmup beside `value`, which is measured behind the clock warm-up below
    el_exact = None
    if not a.exact_warmup and a.shard == "batch":
        el_exact = timed(a.steps)
        warm_run += a.steps
    if not a.exact_warmup and a.shard == "batch":
        # clock warm-up: the GPU reaches its steady clocks after ~0.25 s of load; a short run right behind import / capture otherwise times
        # the ramp (measured: --steps 20 reads 815 steps/s behind 5 warm-up steps and 833 behind 200).  Untimed, reported as warmup_steps_run.
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        run(4 * G, a.epoch)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t_w) / (4 * G)
        extra = int(max(0.0, 0.3 - warm_run * per) / max(per, 1e-6)) // G * G
        extra = min(extra, 2000)
        if dp is not None:                               # every rank must enqueue the SAME number of steps (their collectives pair up)
            extra = int(dp.max_over_ranks(extra)) // G * G
        if extra > 0:
            run(extra, a.epoch)
        warm_run += 4 * G + extra
    el = timed(a.steps)
    loss = stepper.losses()
    rep_rates = [a.steps / el]
    for _ in range(max(a.repeats, 1) - 1):                       # further repeats of the identical timed region (informative; `value` stays the first)
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
        t0r = time.perf_counter()
        run(a.steps, a.epoch)
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
        elr = time.perf_counter() - t0r
        if dp is not None:
            elr = dp.max_over_ranks(elr)
        rep_rates.append(a.steps / elr)

    # random-mask phase (epochs 1..change_epoch) rate, informative
    rnd_rate = None
    if world == 1 and a.shard != "nodes":
        run(max(5, G), 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(max(a.steps // 4, 10), 1)
        torch.cuda.synchronize()
        rnd_rate = max(a.steps // 4, 10) / (time.perf_counter() - t1)

    # per-collective durations (HIP event pair around each RCCL enqueue of three eager steps, on the launch stream) — with the bucketed
    # exchange: the decoder bucket (under the encoder's backward), the encoder bucket and [KL | statistics] behind the chain, the label gather.
    # EVERY rank runs these steps (they hold collectives: a rank that left early would hang the others — ADVICE r04); rank 0 reports.
    coll = {}
    if dp is not None and getattr(dp, "native", None) is not None and a.shard != "nodes":
        from gptst_amd import dist as gdist
        graph_was = stepper.use_graph
        stepper.use_graph = False
        stepper.step(stepper.src, a.epoch)
        torch.cuda.synchronize()
        gdist.COMM_TIMER = []
        for _ in range(3):
            stepper.step(stepper.src, a.epoch)
        torch.cuda.synchronize()
        rec, gdist.COMM_TIMER = gdist.COMM_TIMER, None
        stepper.use_graph = graph_was
        agg = {}
        for what, nb, e0, e1 in rec:
            v = agg.setdefault("%s[%.2f MB]" % (what, nb / 1e6), [0.0, 0])
            v[0] += e0.elapsed_time(e1) * 1e3
            v[1] += 1
        coll["collectives_us"] = {k: round(v[0] / v[1], 1) for k, v in agg.items()}
        coll["collectives_per_step"] = len(rec) // 3
        coll["dp_overlap"] = bool(getattr(stepper, "dp_overlap", False))
        dp.barrier()

    if rank != 0:
        if dp is not None:
            import torch.distributed as dist
            dp.barrier()
            dist.destroy_process_group()
        return
    steps_s = a.steps / el
    strong = a.shard == "nodes" or a.scaling == "strong"
    gbatch = B if a.shard == "nodes" else B * a.gpus
    b32_per_step = gbatch / 32.0 if (a.shard == "batch" and a.scaling == "strong") else (1 if a.shard == "nodes" else a.gpus)
    dims = dict(B=B, T=T, N=N, C=C, HS=args.HS, R=args.num_route)
    out = {
        # whole-job aggregate: per-GPU-batch steps per second summed over the ranks (weak-scaling data parallelism processes
        # n_gpus batches of B per optimizer step); node sharding (strong scaling) processes ONE batch per step
        "metric": "pretrain steps/sec at (B=%d,T=%d,N=%d,C=%d)" % (B, T, N, C),
        "value": steps_s * b32_per_step, "unit": "steps/s",
        "value_definition": ("batches of 32 samples processed per second by the whole job (= optimizer steps/s x global batch %d / 32)" % gbatch)
                            if (a.shard == "batch" and a.scaling == "strong") else
                            ("batches of B=%d processed per second by the whole job (= optimizer steps/s x n_gpus under data parallelism)" % B),
        "optimizer_steps_per_s": steps_s,
        "value_exact_warmup": (a.steps / el_exact * b32_per_step) if el_exact is not None else steps_s * b32_per_step,   # K steps right behind exactly W warm-up steps
        "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "warmup_steps_run": warm_run, "ms_per_step": 1e3 * el / a.steps,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %s-shape synthetic pretrain step, per-GPU B=%d T=%d N=%d C=%d base=%d, epoch %d/300 (%s), "
                               "fwd+loss+bwd+clip+Adam, hipGraph=%s, %s" % (
                                   "BASELINE configs[4] shape (unsharded)" if (a.nodes or a.hidden) else
                                   {"PEMS08": "BASELINE configs[1]", "METR_LA": "BASELINE configs[2] shape", "NYC_TAXI": "BASELINE configs[3] shape"}.get(a.dataset, a.dataset),
                                   a.dataset, B, T, N, C, args.input_base_dim, a.epoch,
                                   "adaptive mask + KL" if a.epoch > args.change_epoch else "random mask",
                                   bool(getattr(stepper, "shard_graph", False)) if a.shard == "nodes" else not a.no_graph,
                                   ("%d full optimiser steps on %d different resident batches per graph replay" % (G, G)) if G > 1 else "one step per replay"),
                   "global_batch": gbatch,
                   "parallelism": ("nodes%d" if a.shard == "nodes" else "dp%d") % a.gpus},
        "samples_per_s": steps_s * gbatch,
        "repeat_values": [r * b32_per_step for r in rep_rates] if len(rep_rates) > 1 else None,
        "median_value": (sorted(rep_rates)[len(rep_rates) // 2] * b32_per_step) if len(rep_rates) > 1 else None,
        "steps_per_s_random_mask_phase": rnd_rate,
        "last_loss": loss[0],
        "handoff_timeouts": _handoff_timeouts(),     # in-launch hand-off waits that expired on rank 0 (0 in a healthy run; an expiry turns the loss NaN)
        # evidence that the job's collectives span the ranks it was started with, and how the step is enqueued
        "rccl_ranks": (group.comm.count() if (a.shard == "nodes" and a.native_comm) else (dp.rccl_ranks() if dp is not None else None)),
        "comm": (None if (dp is None and world == 1) else
                 ("c-abi rccl, captured in the step graph" if (a.native_comm and not getattr(stepper, "_graph_comm_failed", False)) else
                  ("c-abi rccl between graph replays (capturing the collectives failed)" if a.native_comm else "torch.distributed between graph replays"))),
        "graph": bool(getattr(stepper, "shard_graph", False)) if a.shard == "nodes" else not a.no_graph,
        "steps_per_graph_replay": 1 if getattr(stepper, "_group_failed", False) else G,
        "hbm_peak_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2),
        "alloc_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0)),
    }
    # whole-step roofline (64A + side + optimiser bytes; 3 x forward FLOPs)
    A_bytes = 4.0 * B * T * N * C
    bytes_alg = 64 * A_bytes + 4.0 * B * T * N * (8 * args.HS + 4 * (args.input_base_dim + args.HS)) + 28.0 * sum(p.numel() for p in model.parameters())
    d_, ds_, HT, Hm, HS, R = args.embed_dim, args.embed_dim_spa, args.HT, args.HT_Tem, args.HS, args.num_route
    BTN = B * T * N
    F_ht = 2 * BTN * C * C + 4 * B * Hm * T * N * C + 2 * B * T * d_ * C * C + 2 * N * d_ * Hm * T
    F_cap = 4 * BTN * C * C + (2 * R + 3) * 2 * B * T * HS * N * C + 2 * N * d_ * C * C + 4 * B * HT * T * HS * C + 2 * B * T * ds_ * HS * N
    F_mlp = 4 * BTN * C * C + 2 * N * d_ * C * C + 2 * B * T * d_ * C * C + 2 * BTN * C * (args.input_base_dim + HS)
    flops_alg = 3.0 * (8 * F_ht + 4 * F_cap + F_mlp + 4 * BTN * args.input_base_dim * C)
    t_step = el / a.steps
    out["step_roofline"] = {"bytes_alg": bytes_alg, "flops_alg": flops_alg, "t_hbm_us": 1e6 * bytes_alg / HBM_PEAK,
                            "t_mfma_us": 1e6 * flops_alg / MFMA_F32_PEAK, "hbm_frac": (bytes_alg / HBM_PEAK) / t_step,
                            "mfma_frac": (flops_alg / MFMA_F32_PEAK) / t_step}

    if not a.no_kernel_timing and world == 1 and a.shard != "nodes":
        kt = time_kernels(stepper, a.epoch)
        tot = sum(v["total_s"] for v in kt.values())
        ranked = sorted(kt.items(), key=lambda kv: -kv[1]["total_s"])
        A_bytes = 4.0 * B * T * N * C

        def kernel_roofline(dn, dt, dv):
            """SURVEY 8(d) protocol for one kernel family: achieved = ALGORITHMIC bytes (8d per-layer figure; flops likewise) / measured duration, against
            the roof that binds the kernel (the larger of its two floor times); the operand-byte figure (every operand once) is kept beside it.
            Duration: the average over the launches of the three timed eager steps (HIP event pair around every launch on the launch stream) — i.e. IN
            the chain, behind its real producer; the event pair adds 2-5 us of dispatch gap per launch, so `avg_us_graph` — the same kernel's rocprofv3
            average under hipGraph replay from the committed trace, present when that trace was taken on these kernel sources — stands beside it and
            `frac_graph` is the fraction on it."""
            fl = kernel_flops(dn, dt, dims)
            b8d = ALG_8D_A.get(dn, None)
            if dn == "gptst_hypertem_chain_fwd":               # 2A per hyperTem layer of the chain
                b8d = 2.0 * int(dt.split()[-1][1:])
            b8d = A_bytes * b8d if b8d is not None else float(dv["bytes"])
            t_h, t_m = b8d / HBM_PEAK, fl / MFMA_F32_PEAK
            hint = {"mode0": "[384,", "mode1": "[170,", "mode2": "[1,"}.get(dt.split()[0]) if dt else None
            ug = graph_avg_us(dn, dt, hint)
            if t_h >= t_m:
                rf = dict(bound="hbm", achieved=b8d / dv["avg_s"] / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s")
                fg = (b8d / (ug * 1e-6)) / HBM_PEAK if ug else None
            else:
                rf = dict(bound="mfma", achieved=fl / dv["avg_s"] / 1e12, peak=MFMA_F32_PEAK / 1e12, unit="TFLOP/s")
                fg = (fl / (ug * 1e-6)) / MFMA_F32_PEAK if ug else None
            rf["frac"] = rf["achieved"] / rf["peak"]
            rf["traffic"], rf["traffic_kernel"] = pmc_traffic(dn, dt, hint)
            rf.update(kernel="%s[%s]" % (dn, dt), avg_us=1e6 * dv["avg_s"], avg_us_graph=ug, frac_graph=fg,
                      us_per_step=1e6 * dv["total_s"], us_per_step_graph=(ug * dv["launches"] if ug else None),
                      timing="HIP event pair around every launch of the kernel inside three eager steps (on the launch stream), averaged",
                      launches_per_step=dv["launches"],
                      alg_bytes_8d=b8d, alg_flops_per_launch=fl, frac_8d=(b8d / dv["avg_s"]) / HBM_PEAK, mfma_frac=(fl / dv["avg_s"]) / MFMA_F32_PEAK,
                      operand_bytes_per_launch=dv["bytes"], frac_operand_bytes=(dv["bytes"] / dv["avg_s"]) / HBM_PEAK,
                      share_of_step_kernel_time=dv["total_s"] / tot)
            return rf
        # The two heaviest kernel families are within 8 % of each other (routing backward ~178 us / step, hyperTem backward pair ~165), and eager event
        # pairs inflate the one with more launches: BOTH are reported (`top2`), the first — by eager time, as in every earlier round — is `roofline`.
        (dn, dt), dv = ranked[0]
        rf = kernel_roofline(dn, dt, dv)
        precise = retime_kernel(dn, dt)
        rf["avg_us_back_to_back"] = 1e6 * precise if precise is not None else None
        rf["kernel_src_sha"] = kernel_source_hash()
        rf["top2"] = [{k: v for k, v in kernel_roofline(n_, t_, v_).items() if k not in ("timing",)} for (n_, t_), v_ in ranked[:2]]
        out["roofline"] = rf
        top = ranked[:12]
        out["kernel_breakdown_us_per_step"] = {"%s[%s]" % k: round(1e6 * v["total_s"], 1) for k, v in top}
        out["kernel_time_sum_us_per_step_eager"] = round(1e6 * tot, 1)
    out.update(coll)
    if world == 1 and a.shard != "nodes" and (a.path == "module" or not a.no_module_path):
        stepper = None
        torch.cuda.empty_cache()
        out["module_path"] = module_path(args, B, a.epoch, dev, steps=a.steps if a.path == "module" else 40)
        torch.cuda.empty_cache()
        out["module_path_clipadam"] = module_path(args, B, a.epoch, dev, steps=a.steps if a.path == "module" else 40, clip_adam=True)
        if a.path == "module":
            out["value"], out["ms_per_step"] = out["module_path"]["steps_per_s"], out["module_path"]["ms_per_step"]
            out["optimizer_steps_per_s"] = out["value"]
            out["config"]["workload"] += "; PATH = module (reference-style loop over the drop-in nn.Module, not the fused step)"
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, B)
    print(json.dumps(out), flush=True)
    if dp is not None:
        import torch.distributed as dist
        dp.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
