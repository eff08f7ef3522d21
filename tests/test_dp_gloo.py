"""world_size-2 data-parallel reduction on CPU (gloo): the packed [sum-loss gradient | statistics] all-reduce of dist.py plus the
optimiser-side scaling reproduces the gradient of the global-batch loss.  Local gradients come from the oracle (the checker) —
no HIP compute happens here; the same DataParallel class runs over RCCL on the GPUs."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import free_port  # noqa: E402
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _local_buffer(rank, world, args, sd, src_all, masks_all, keys, nA_keys):
    """[sum-loss gradient (path-A params first) | stats] of this rank's slice, from the oracle."""
    from gptst_amd import synth
    from oracle import gptst_oracle as O
    B = src_all.shape[0] // world
    src, mask = src_all[rank * B:(rank + 1) * B], masks_all[rank * B:(rank + 1) * B]
    st = O.Stepper(sd, args, synth.SCALER_MEAN, synth.SCALER_STD)
    outs, _ = O.forward_pretrain(st.sd, args, src, 1, forced_mask=mask)
    out, m = outs[0], outs[2]
    p = (out * synth.SCALER_STD + synth.SCALER_MEAN) * m
    y = (src[..., :1] * synth.SCALER_STD + synth.SCALER_MEAN) * m
    keep = y > args.mape_thresh
    lsum = (torch.abs(y - p) * keep).sum()
    lsum.backward()
    g = torch.cat([(st.sd[k].grad if st.sd[k].grad is not None else torch.zeros_like(st.sd[k])).reshape(-1) for k in keys])
    stats = torch.zeros(8)
    stats[0], stats[1] = float(lsum), float(keep.sum())
    return torch.cat([g, stats])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from gptst_amd import synth
    from gptst_amd.config import make_args
    from gptst_amd.dist import DataParallel
    from oracle import gptst_oracle as O
    args = make_args("PEMS08", num_nodes=12, embed_dim=4, HS=4, HT=4, scaler_zeros=synth.scaler_zeros())
    sd = O.init_state_dict(args, 2)
    keys = [k for k in sd if not k.endswith("mask_template")]
    src_all = synth.make_batch(4, 12, 12, 1, seed=3)
    masks_all = (torch.rand(4, 12, 12, 1, generator=torch.Generator().manual_seed(5)) > 0.25).long()
    buf = _local_buffer(rank, world, args, sd, src_all, masks_all, keys, None)
    dp = DataParallel("gloo")
    dp.allreduce_(buf)
    n = buf.numel() - 8
    g = buf[:n] / max(float(buf[n + 1]), 1.0)
    if rank == 0:
        q.put((g, float(buf[n] / buf[n + 1])))
    dp.barrier()


def test_dp_allreduce_equals_global_batch_gradient():
    sys.path.insert(0, ROOT)
    from gptst_amd import synth
    from gptst_amd.config import make_args
    from oracle import gptst_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g, loss = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: global batch of 4 with the mean loss
    args = make_args("PEMS08", num_nodes=12, embed_dim=4, HS=4, HT=4, scaler_zeros=synth.scaler_zeros())
    sd = O.init_state_dict(args, 2)
    keys = [k for k in sd if not k.endswith("mask_template")]
    src_all = synth.make_batch(4, 12, 12, 1, seed=3)
    masks_all = (torch.rand(4, 12, 12, 1, generator=torch.Generator().manual_seed(5)) > 0.25).long()
    st = O.Stepper(sd, args, synth.SCALER_MEAN, synth.SCALER_STD)
    outs, _ = O.forward_pretrain(st.sd, args, src_all, 1, forced_mask=masks_all)
    lf = O.mae_loss(outs[0], src_all[..., :1], outs[2], synth.SCALER_MEAN, synth.SCALER_STD, args.mape_thresh)
    lf.backward()
    ref = torch.cat([(st.sd[k].grad if st.sd[k].grad is not None else torch.zeros_like(st.sd[k])).reshape(-1) for k in keys])
    assert abs(loss - float(lf)) < 1e-5 * abs(float(lf))
    torch.testing.assert_close(g, ref, rtol=2e-4, atol=1e-6)


def test_combine_local_gradients_matches_formula():
    from gptst_amd.dist import combine_local_gradients
    a = torch.arange(20, dtype=torch.float32); b = torch.ones(20)
    a[-8:] = torch.tensor([10., 4., 1., 0, 0, 0, 0, 0]); b[-8:] = torch.tensor([6., 2., 3., 0, 0, 0, 0, 0])
    g, stats = combine_local_gradients([a, b], nA=8, nB=4)
    assert float(stats[1]) == 6.0 and float(stats[0]) == 16.0
    torch.testing.assert_close(g[:8], (a[:8] + b[:8]) / 6.0)
    torch.testing.assert_close(g[8:], a[8:12] + b[8:12])


def _worker_labels(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from gptst_amd.dist import DataParallel
    dp = DataParallel("gloo")
    M = 2 * 12 * 5                                                       # this rank's B*T*N cells
    g = torch.Generator().manual_seed(100 + rank)
    label = torch.randint(0, 4, (M,), generator=g, dtype=torch.int32)
    counts = torch.bincount(label, minlength=4).to(torch.int32)
    lab_g = dp.gather_labels(label)
    dp.sum_counts_(counts)
    mask_g = torch.arange(world * M * 2, dtype=torch.float32)            # any batch-major global vector (base = 2)
    mine = dp.rows_of(mask_g, M * 2)
    q.put((rank, lab_g.tolist(), counts.tolist(), mine.tolist()))        # plain lists: the worker may exit before the parent reads
    dp.barrier()


def test_dp_label_exchange_protocol():
    """Adaptive-mask phase under DP (dist.py): labels are concatenated in rank order (= batch-major order of the global batch),
    class counts are summed, and a rank's rows of a global vector are its contiguous slice."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_labels, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    M = 2 * 12 * 5
    labs = [torch.randint(0, 4, (M,), generator=torch.Generator().manual_seed(100 + r), dtype=torch.int32) for r in range(2)]
    want_lab = torch.cat(labs)
    want_cnt = torch.bincount(want_lab, minlength=4).to(torch.int32)
    for r, lab_g, counts, mine in got:
        assert lab_g == want_lab.tolist() and counts == want_cnt.tolist()
        assert mine == torch.arange(2 * M * 2, dtype=torch.float32)[r * M * 2:(r + 1) * M * 2].tolist()


def test_mesh_groups_layout():
    """SURVEY 8(e) "Combination": world = dp x shard, rank = dp_index * n_shard + shard_index; rows are the node-shard communicators, columns
    the data-parallel ones; every rank is in exactly one of each."""
    from gptst_amd.dist import mesh_groups, mesh_shape
    assert mesh_shape(8, 4) == (2, 4)
    rows, cols = mesh_groups(8, 4)
    assert rows == [[0, 1, 2, 3], [4, 5, 6, 7]] and cols == [[0, 4], [1, 5], [2, 6], [3, 7]]
    for world, ns in ((8, 1), (8, 8), (6, 2)):
        rows, cols = mesh_groups(world, ns)
        for r in range(world):
            assert sum(r in g for g in rows) == 1 and sum(r in g for g in cols) == 1


def test_bucketed_allreduce_equals_the_single_allreduce():
    """r04: the packed [gradient | statistics] buffer leaves in three pieces — the decoder's bucket early (under the encoder's backward), then
    [encoder] and [KL path | never trained | statistics] — which must reduce to exactly what the one all-reduce gives, and the decoder's
    trained parameters must be ONE contiguous range of the flat buffer that ends where the reconstruction-path segment ends."""
    from gptst_amd.config import make_args
    from gptst_amd.model import GPTST_Model
    from gptst_amd.step import PretrainStep
    args = make_args("PEMS08", num_nodes=20, embed_dim=4)
    m = GPTST_Model(args)
    lo, hi = PretrainStep._decoder_bucket(m)
    assert 0 < lo < hi == m.nA
    inside = {k for k, o in m._offs.items() if lo <= o < hi}
    assert inside and all(k.startswith("decoder.") for k in inside)
    assert {k for k in m._offs if k.startswith("decoder.") and not k.startswith("decoder.time_feature")} == inside
    n = m.flat.numel()
    g = torch.Generator().manual_seed(3)
    bufs = [torch.randn(n + 8, generator=g) for _ in range(3)]
    whole = torch.stack(bufs).sum(0)
    pieces = torch.zeros(n + 8)
    for a, b in ((lo, hi), (0, lo), (hi, n + 8)):
        pieces[a:b] = torch.stack([t[a:b] for t in bufs]).sum(0)
    assert torch.equal(whole, pieces)
