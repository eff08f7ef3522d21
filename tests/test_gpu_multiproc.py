"""The multi-process entry of bench.py (what the driver launches for N > 1) exercised on ONE GPU: two ranks share cuda:0 and talk
over gloo instead of RCCL.  Covers process-group setup, the [gradient | statistics] all-reduce, the label exchange of the
adaptive-mask phase, the node-sharded mode, max-over-ranks timing and the single JSON line on rank 0."""
import json
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import free_port  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, GPTST_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "4"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,scaling,par", [([], "weak", "dp2"), (["--epoch", "1"], "weak", "dp2"), (["--shard", "nodes"], "strong", "nodes2")])
def test_bench_two_ranks(extra, scaling, par):
    out = _run(extra, free_port())
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["config"]["parallelism"] == par
    assert out["value"] > 0 and out["steps"] == 4 and out["higher_is_better"] is True
    assert out["handoff_timeouts"] == 0, "a bounded in-launch hand-off wait expired (two processes share this GPU)"
    assert out["last_loss"] == out["last_loss"] and out["last_loss"] > 0          # finite


def _run_plain(args, env_extra, timeout=900):
    """`python bench.py ...` with NO launcher and no WORLD_SIZE in the environment (the shape of the driver's N = 1 command)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` spawns its two ranks itself (torch.distributed.run on 127.0.0.1) and still prints ONE JSON line."""
    out = _run_plain(["--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "4"], dict(GPTST_DIST_BACKEND="gloo"))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "dp2" and out["value"] > 0


def test_bench_strong_scaling_splits_global_batch():
    out = _run_plain(["--gpus", "2", "--steps", "4", "--warmup", "1", "--scaling", "strong", "--global-batch", "8"],
                     dict(GPTST_DIST_BACKEND="gloo"))
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 8 and "B=4 " in out["config"]["workload"]
    assert abs(out["value"] - out["optimizer_steps_per_s"] * 8 / 32.0) < 1e-6 * out["value"]


def test_bench_rccl_backend():
    """The RCCL ("nccl") process group, all-reduce of [gradient | statistics], label all-gather and count all-reduce on real
    hardware: two ranks when the box has two GPUs, otherwise ONE rank forced through the data-parallel path."""
    import torch
    if torch.cuda.device_count() >= 2:
        out = _run_plain(["--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "4"], {})
        assert out["n_gpus"] == 2
    else:
        out = _run_plain(["--gpus", "1", "--steps", "6", "--warmup", "2", "--batch", "4", "--no-cpu-baseline", "--no-kernel-timing"],
                         dict(GPTST_FORCE_DP="1", GPTST_DIST_BACKEND="nccl"))
        assert out["n_gpus"] == 1
    assert out["value"] > 0 and out["last_loss"] == out["last_loss"] and out["last_loss"] > 0
    # the collectives ran on the C-ABI communicator inside the step's hipGraph, and RCCL itself reports the rank count of the job
    assert out["rccl_ranks"] == out["n_gpus"] and out["graph"] is True and "c-abi" in out["comm"]


def test_bench_dp_torch_comm_fallback():
    """--torch-comm: torch.distributed (RCCL) collectives between graph replays, as before round 3 — still one line, still counts its ranks."""
    out = _run_plain(["--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "4", "--no-cpu-baseline", "--no-kernel-timing", "--torch-comm"],
                     dict(GPTST_FORCE_DP="1", GPTST_DIST_BACKEND="nccl"))
    assert out["value"] > 0 and out["rccl_ranks"] == 1 and "torch.distributed" in out["comm"]


def test_dp_step_in_one_graph_equals_the_plain_step():
    """Data parallel on the capturable communicator (one rank): label gather, gradient all-reduce and optimiser are nodes of the step's ONE
    hipGraph; a step sequence crossing both mask phases gives the same losses and weights as the plain single-GPU stepper."""
    import torch
    import torch.distributed as dist
    from gptst_amd import synth
    from gptst_amd.config import make_args
    from gptst_amd.dist import DataParallel
    from gptst_amd.model import GPTST_Model
    from gptst_amd.step import PretrainStep
    from oracle import gptst_oracle as O
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    args = make_args("PEMS08", num_nodes=24, embed_dim=8, HS=6, HT=8, scaler_zeros=synth.scaler_zeros(), epochs=30, change_epoch=3)
    sd = O.init_state_dict(args, 3)
    B, M = 4, 4 * 12 * 24
    src = synth.make_batch(B, 12, 24, 1, seed=5).to("cuda:0")
    dp = DataParallel("nccl", native=True)
    try:
        assert dp.capturable and dp.rccl_ranks() == 1
        res = []
        for d, overlap in ((None, "0"), (dp, "0"), (dp, "1"), (dp, "2")):
            # overlap "1" / "2" (r04): the decoder's gradient bucket is all-reduced (2: also reduced) on a forked branch under the encoder's
            # backward, the rest ([encoder] and [KL path | statistics]) behind the chain — three collectives in the graph instead of one
            os.environ.update(GPTST_FORCE_DP="1", GPTST_DP_OVERLAP=overlap)
            model = GPTST_Model(args)
            model.load_state_dict(sd)
            model = model.to("cuda:0")
            st = PretrainStep(model, args, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=True, dp=d, seed=7, deterministic=True)
            assert st.dp_overlap == (d is not None and overlap != "0")
            assert 0 < st.dec_lo < st.dec_hi == model.nA
            losses = []
            for i, epoch in enumerate((1, 1, 20, 20, 20)):
                kw = (dict(noise=synth.make_noise(M, 10 + i).to("cuda:0")) if epoch == 1 else
                      dict(noise_a=synth.make_noise(M, 20 + i).to("cuda:0"), noise_r=synth.make_noise(M, 30 + i).to("cuda:0"),
                           list_c=synth.class_order(6, 40 + i)))
                st.step(src, epoch, **kw)
                losses.append(st.losses())
            if d is not None:
                assert all(g2 is None for _, g2 in st.graphs.values()), "one graph per phase, collectives inside"
            res.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
        for other in res[1:]:
            for a, b in zip(res[0][0], other[0]):
                assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0]), (a, b)
            worst = max(float((res[0][1][k].float() - other[1][k].float()).abs().max()) for k in res[0][1])
            assert worst < 2e-6, worst              # (fixed-order reductions on both sides: measured 0 .. 1e-6 over five Adam steps at lr 3e-3)
    finally:
        os.environ.pop("GPTST_FORCE_DP", None); os.environ.pop("GPTST_DP_OVERLAP", None)
        dp.native.close()
        dist.destroy_process_group()


def test_bench_stdout_is_one_line_with_rccl_banner_enabled():
    """The GPU boxes export NCCL_DEBUG=VERSION: RCCL then prints a version banner to STDOUT, which (C stdio, pipe) lands behind the JSON
    line.  The bench drops the variable for itself and its ranks: the RCCL path still prints exactly one line, and it is the JSON."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GPTST_FORCE_DP="1", GPTST_DIST_BACKEND="nccl", NCCL_DEBUG="VERSION")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "4", "--no-cpu-baseline",
                        "--no-kernel-timing"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["value"] > 0, r.stdout[-2000:]


def test_native_comm_allreduce_inside_a_graph():
    """C-ABI communication entry points (gptst_comm_* / gptst_allreduce_f32: RCCL bound at run time) with one rank: the all-reduce is an
    ordinary stream enqueue — eager and captured in a hipGraph together with a kernel — and leaves a 1-rank sum unchanged."""
    import torch
    from gptst_amd.dist import NativeComm
    comm = NativeComm(rank=0, world=1)
    try:
        x = torch.arange(1 << 20, device="cuda:0", dtype=torch.float32)
        ref = x.clone()
        comm.allreduce_(x)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            comm.allreduce_(x)                                     # warm-up outside capture
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            x.mul_(2.0)
            comm.allreduce_(x)
        g.replay(); g.replay()
        torch.cuda.synchronize()
        assert torch.equal(x, ref * 4)                               # two replays (capture records, it does not execute)
        # the label all-gather of a data-parallel global batch (gptst_allgather_i32): eager and inside a graph; with one rank a copy
        from gptst_amd.dist import DataParallel
        lab = torch.randint(0, 10, (65280,), device="cuda:0", dtype=torch.int32)
        out = torch.full((65280,), -1, device="cuda:0", dtype=torch.int32)
        assert comm.allgather_i32(lab, out)
        torch.cuda.synchronize()
        assert torch.equal(out, lab)
        out.fill_(-1)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            comm.allgather_i32(lab, out)
        g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, lab)
        comm._no_allgather = True                                     # a library without ncclAllGather: the slot all-reduce takes over
        dp = DataParallel.__new__(DataParallel)
        dp.native, dp.world, dp.rank, dp._slots = comm, 1, 0, None
        assert torch.equal(dp.gather_labels(lab), lab)
    finally:
        comm.close()


def test_two_native_communicators_coexist():
    """r04: communicators are handles (gptst_comm_init -> *comm_out), so one process can hold the row AND the column communicator of a
    data-parallel x node-shard mesh.  With one rank each: both serve collectives on the same stream, independently destroyed."""
    import torch
    from gptst_amd.dist import NativeComm, mesh_comms
    a, b = NativeComm(rank=0, world=1), NativeComm(rank=0, world=1)
    try:
        assert a.h.value != b.h.value and a.count() == 1 and b.count() == 1
        x = torch.arange(4096, device="cuda:0", dtype=torch.float32)
        ref = x.clone()
        a.allreduce_(x); b.allreduce_(x); a.allreduce_(x)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        a.close()
        b.allreduce_(x)                                                # b outlives a
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
    finally:
        a.close(); b.close()
    shard, dp = mesh_comms(1, rank=0, world=1)                         # the 1 x 1 mesh: two communicators of one rank
    try:
        assert shard.world == 1 and dp.world == 1 and shard.h.value != dp.h.value
    finally:
        shard.close(); dp.close()


def test_trainer_keeps_the_tail_of_a_data_parallel_epoch(tmp_path):
    """r04 (verdict 5e): two ranks (gloo, one GPU) train two epochs — random-mask and adaptive phase — of a series with an ODD number of full
    batches and a ragged last one.  Every batch of the epoch is stepped on: whole groups through the captured step, the left-over full batch and the
    ragged batch as padded rounds (the rank without a batch contributes weight 0); both ranks end with bit-identical weights."""
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_tail_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, GPTST_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["full"] % 2 == 1 and out["n"] % 8 != 0, out                 # the epoch HAS a left-over full batch and a ragged one
    assert out["tail_rounds"] == [1, 1] and out["nb"] == out["full"] // 2 + 2
    assert out["tA"] == 2 * out["nb"], out                                  # every round of both epochs was an optimiser step
    assert out["same_weights"], "the replicas diverged"
    assert all(l == l and l > 0 for l in out["losses"]), out



def test_trainer_train_end_to_end_under_data_parallelism(tmp_path):
    """Trainer.train() closes with test() over the epoch's batches, which under data parallelism include the (batch, weight)
    tuples of the padded tail rounds.  Two gloo ranks on one GPU run train() to the end: the evaluation completes and counts every sample of the
    training set exactly once (padding — weight 0 — is skipped, the metric sums are all-reduced)."""
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_tail_worker.py"), str(tmp_path), "train"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, GPTST_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["trained"] and out["same_weights"], out
    assert out["eval_samples"] == out["n"], out
