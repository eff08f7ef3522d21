"""Node-sharded step (gpt-st_amd/shard.py): two ranks, emulated by two threads on this GPU with a barrier-based
all-reduce, must reproduce the unsharded step — same masks (bit exact), same losses, same parameter update — in the random-mask
phase and in the adaptive-mask + KL phase."""
import threading

import pytest
import torch

from gptst_amd import synth
from gptst_amd.config import make_args
from oracle import gptst_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


SMALL = dict(embed_dim=8, HS=5, HT=6)
CONFIG4 = dict(hidden_dim=128)              # BASELINE configs[4] dims: C = 128, HS = 10, d = 16 (N per shard 256 / 512 below)


def _args(n, over=SMALL):
    return make_args("PEMS08", num_nodes=n, num_route=2, scaler_zeros=synth.scaler_zeros(), epochs=30, change_epoch=3, **over)


def _run_sharded(W, N, B, over, steps, srcs, noise, list_c, sd):
    """W ranks emulated by threads on this GPU -> per rank (losses, global masks, state_dict, stats snapshots)."""
    from gptst_amd import ops
    from gptst_amd.model import GPTST_Model
    from gptst_amd.shard import ShardedPretrainStep, ThreadNodeGroup, shard_state_dict
    Nl = N // W
    shared = ThreadNodeGroup.Shared(W)
    ops.CALL_LOCK = threading.Lock()
    out, errs = [None] * W, []

    def rank_main(r):
        try:
            args_l = _args(Nl, over)
            m = GPTST_Model(args_l); m.load_state_dict(shard_state_dict(sd, r * Nl, (r + 1) * Nl)); m = m.to(DEV)
            s = ShardedPretrainStep(m, args_l, N, ThreadNodeGroup(r, shared), synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B)
            losses, masks, stats = [], [], []
            for (epoch, _), src, (n0, na, nr) in zip(steps, srcs, noise):
                s.step(src[:, :, r * Nl:(r + 1) * Nl].contiguous(), epoch, noise=n0, noise_a=na, noise_r=nr, list_c=list_c)
                losses.append(s.losses()); masks.append(s.last_mask_global.clone()); stats.append(s.stats_out.cpu().clone())
            out[r] = (losses, masks, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, stats)
        except BaseException as e:              # noqa: BLE001 - surface the failure in the main thread
            errs.append(e)
            shared.barrier.abort()

    try:
        ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(600)
    finally:
        ops.CALL_LOCK = None
    assert not errs, errs
    return out


@pytest.mark.parametrize("W,N,B,over", [(2, 40, 2, SMALL), (4, 40, 2, SMALL), (2, 1024, 1, CONFIG4), (4, 1024, 1, CONFIG4)],
                         ids=["w2_n40", "w4_n40", "w2_n1024_c128", "w4_n1024_c128"])
def test_node_shards_equal_unsharded_step(W, N, B, over, parity, monkeypatch):
    """Both steppers run with fixed-order reductions (GPTST_DETERMINISTIC): with atomics the three-step weight comparison below changed from run
    to run (r04: 5e-4 .. 1.7e-3 on the same tensors) and a bound on it was a bound on luck.
    (w*_c128: BASELINE configs[4] — C = 128 through reduce_nodes on 512 / 256 nodes per shard; the unsharded step is also held
    against the CPU oracle there, with the 5-D tensor of the reference not materialised.)"""
    from gptst_amd.model import GPTST_Model
    from gptst_amd.shard import unshard_state_dicts
    from gptst_amd.step import PretrainStep
    monkeypatch.setenv("GPTST_DETERMINISTIC", "1")
    args_g = _args(N, over)
    sd = O.init_state_dict(args_g, 5)
    Mg = B * 12 * N
    HS = args_g.HS
    steps = [(1, 0), (20, 1), (25, 2)]                     # (epoch, seed): random phase, then adaptive + KL twice
    srcs = [synth.make_batch(B, 12, N, 1, seed=40 + s).to(DEV) for _, s in steps]
    noise = [tuple(synth.make_noise(Mg, 10 * s + i).to(DEV) for i in range(3)) for _, s in steps]
    list_c = synth.class_order(HS, 9)

    # ---- unsharded reference ----
    model = GPTST_Model(args_g); model.load_state_dict(sd); model = model.to(DEV)
    st = PretrainStep(model, args_g, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=False)
    ref_loss, ref_mask = [], []
    for (epoch, _), src, (n0, na, nr) in zip(steps, srcs, noise):
        st.step(src, epoch, noise=n0, noise_a=na, noise_r=nr, list_c=list_c)
        ref_loss.append(st.losses()); ref_mask.append(st.last_mask.clone())
    ref_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    if over is CONFIG4:                                  # the unsharded HIP step against the oracle at this shape
        ora = O.Stepper(sd, args_g, synth.SCALER_MEAN, synth.SCALER_STD, materialize_5d=False)
        for i, ((epoch, _), src, (n0, na, nr)) in enumerate(zip(steps, srcs, noise)):
            kw = dict(noise=n0.cpu()) if epoch <= args_g.change_epoch else dict(noise_a=na.cpu(), noise_r=nr.cpu(), list_c=list_c)
            r = ora.step(src.cpu(), epoch, **kw)
            vis = ref_mask[i].cpu().view(B, 12, N, 1)
            if not torch.equal((1 - vis).long(), r[3][2]):             # an fp32-level argmax flip of the guide: same budget, few cells
                assert int((vis == 0).sum()) == int(r[3][2].sum()) and float(((1 - vis).long() == r[3][2]).float().mean()) > 0.97
                break
            e = abs(ref_loss[i][1] - r[1]) / abs(r[1])
            parity("oracle_flow_loss_step%d" % i, e)
            assert e < 2e-4, (i, ref_loss[i], r[:3])

    out = _run_sharded(W, N, B, over, steps, srcs, noise, list_c, sd)
    got_sd = unshard_state_dicts([out[r][2] for r in range(W)])
    for i in range(len(steps)):
        for r in range(W):
            assert torch.equal(out[r][1][i], ref_mask[i]), "global mask differs at step %d on rank %d" % (i, r)
            for a, b in zip(out[r][0][i], ref_loss[i]):
                assert abs(a - b) <= 2e-4 * max(abs(b), 1e-3), (i, r, out[r][0][i], ref_loss[i])
    worst = worst_raw = 0.0
    for k, v in ref_sd.items():
        if not v.dtype.is_floating_point:
            continue
        upd = v - sd[k]
        d = (got_sd[k] - v).abs().flatten()
        raw = float(d.norm() / upd.norm().clamp_min(1e-6))
        worst_raw = max(worst_raw, raw)
        # Adam turns a round-off-level gradient into a move of up to +-lr per step (m / (sqrt(v) + eps) with |g| ~ eps follows the gradient's
        # round-off, sign included).  cap*.t_adj has such elements — its gradient is a difference of products summed over the nodes, in
        # another order on a shard and, the reductions using atomics, in another order from run to run: over eleven runs of w4_n1024_c128 the
        # update error of decoder cap1.t_adj was 5.2e-4 .. 1.7e-3 of the tensor's update norm (r03: 5.2e-4; the r04 Adam fix moved the
        # trajectory), sometimes concentrated in four elements, sometimes spread.  That is not a sharding error, so t_adj gets the bound its
        # noise needs (4e-3; a wrong reduction is >= 1e-1) and no element may be off by more than one step's flip; every other tensor 1.5e-3 (measured 6.0e-4 .. 7.3e-4 with fixed-order reductions).
        bound = 4e-3 if k.endswith(".t_adj") else 1.5e-3
        assert float(d.max()) <= 2.5 * args_g.lr_init, "%s: element update off by %.3e (lr %.1e)" % (k, float(d.max()), args_g.lr_init)
        err = raw
        worst = max(worst, err if not k.endswith(".t_adj") else 0.0)
        assert err < bound, "%s: update differs, rel-L2 of the update error %.3e" % (k, err)
    parity("update_rel_l2_worst", worst)
    parity("update_rel_l2_worst_incl_t_adj", worst_raw)
    print("worst relative update error %.2e" % worst)


def test_config4_full_size_properties(parity):
    """BASELINE configs[4] at full size (N = 4096, C = 128, B = 2), 4 node shards of 1024 against the unsharded step: exact mask
    budget, finite losses that agree, and the same global gradient norm (size-independent properties; the unsharded step itself is checked
    against the fp64 oracle at this size by tests/test_gpu_shapes.py::test_model_vs_oracle[c5_full_n4096_c128])."""
    from gptst_amd.model import GPTST_Model
    from gptst_amd.step import PretrainStep
    W, N, B = 4, 4096, 2
    args_g = _args(N, CONFIG4)
    sd = O.init_state_dict(args_g, 6)
    Mg = B * 12 * N
    steps = [(1, 0), (20, 1)]
    srcs = [synth.make_batch(B, 12, N, 1, seed=50 + s).to(DEV) for _, s in steps]
    noise = [tuple(synth.make_noise(Mg, 20 * s + i + 1).to(DEV) for i in range(3)) for _, s in steps]
    list_c = synth.class_order(args_g.HS, 4)
    model = GPTST_Model(args_g); model.load_state_dict(sd); model = model.to(DEV)
    st = PretrainStep(model, args_g, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=False)
    ref = []
    for (epoch, _), src, (n0, na, nr) in zip(steps, srcs, noise):
        st.step(src, epoch, noise=n0, noise_a=na, noise_r=nr, list_c=list_c)
        ref.append((st.losses(), st.last_mask.clone(), st.stats_out.cpu().clone()))
    del st, model
    torch.cuda.empty_cache()
    out = _run_sharded(W, N, B, CONFIG4, steps, srcs, noise, list_c, sd)
    budget = int(int(Mg * args_g.mask_ratio))
    for i, (epoch, _) in enumerate(steps):
        (loss, lf, ls), mask, stats = ref[i]
        assert int((mask == 0).sum()) == budget, "mask budget"                    # GPTST.py:318 / :351-353,:388,:399
        assert all(v == v and abs(v) < 1e6 for v in (loss, lf, ls))
        for r in range(W):
            assert torch.equal(out[r][1][i], mask), (i, r)
            e = abs(out[r][0][i][0] - loss) / abs(loss)
            parity("loss_sharded_vs_unsharded", e)
            assert e < 2e-4, (i, r, out[r][0][i], loss)
            gn, gn_ref = float(out[r][3][i][4]) ** 0.5, float(stats[4]) ** 0.5        # stats[4] = total sum g^2 seen by the optimiser
            parity("gradnorm_sharded_vs_unsharded", abs(gn - gn_ref) / gn_ref)
            assert abs(gn - gn_ref) < 1e-3 * gn_ref, (i, r, gn, gn_ref)


@pytest.mark.parametrize("kind", ["dist_world1", "native_comm"])
def test_sharded_step_in_one_hipgraph(kind, parity):
    """A group whose collectives are stream-ordered device work lets the whole node-sharded step — kernels and collectives — be ONE
    hipGraph per phase: world = 1 (collectives are copies) and the C-ABI communicator (RCCL enqueues on the launch stream, here with one
    rank).  Graph replays reproduce the eager sharded steps: same masks, same losses, same parameters (up to float-atomic order)."""
    from gptst_amd.model import GPTST_Model
    from gptst_amd.shard import DistNodeGroup, NativeNodeGroup, ShardedPretrainStep
    N, B = 40, 2
    args = _args(N)
    sd = O.init_state_dict(args, 8)
    Mg = B * 12 * N
    steps = [(1, 0), (2, 1), (20, 2), (25, 3)]
    srcs = [synth.make_batch(B, 12, N, 1, seed=60 + s).to(DEV) for _, s in steps]
    noise = [tuple(synth.make_noise(Mg, 30 * s + i).to(DEV) for i in range(3)) for _, s in steps]
    list_c = synth.class_order(args.HS, 3)
    comm = None
    if kind == "native_comm":
        from gptst_amd.dist import NativeComm
        comm = NativeComm(rank=0, world=1)
    try:
        res = []
        for use_graph in (False, True):
            group = NativeNodeGroup(comm) if comm is not None else DistNodeGroup(0, 1)
            m = GPTST_Model(args); m.load_state_dict(sd); m = m.to(DEV)
            st = ShardedPretrainStep(m, args, N, group, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=use_graph)
            assert st.shard_graph == use_graph
            losses, masks = [], []
            for (epoch, _), src, (n0, na, nr) in zip(steps, srcs, noise):
                st.step(src, epoch, noise=n0, noise_a=na, noise_r=nr, list_c=list_c)
                losses.append(st.losses()); masks.append(st.last_mask_global.clone())
            res.append((losses, masks, m.flat.detach().clone()))
    finally:
        if comm is not None:
            comm.close()
    (le, me, pe), (lg, mg, pg) = res
    for i in range(len(steps)):
        assert torch.equal(me[i], mg[i]), "mask differs at step %d" % i
        for a, b in zip(le[i], lg[i]):
            assert abs(a - b) <= 2e-4 * max(abs(b), 1e-3), (i, le[i], lg[i])
    err = float((pe - pg).norm() / pe.norm())
    parity("param_rel_l2_graph_vs_eager", err)
    assert err < 1e-3


def test_sharded_step_recovers_from_a_lost_handoff():
    """the node-sharded stepper runs the fused hand-off launches too; an expiry on record makes the optimiser skip, losses() takes the
    step back and repeats it on the launches without hand-offs (the base class's recovery: the shard's step() now leaves what that needs)."""
    from gptst_amd import _C
    from gptst_amd.model import GPTST_Model
    from gptst_amd.shard import DistNodeGroup, ShardedPretrainStep
    N, B = 40, 2
    args = _args(N)
    sd = O.init_state_dict(args, 8)
    srcs = [synth.make_batch(B, 12, N, 1, seed=80 + s).to(DEV) for s in range(3)]
    orders = [synth.class_order(args.HS, 5 + s) for s in range(3)]
    res = []
    try:
        for lose in (False, True):
            m = GPTST_Model(args); m.load_state_dict(sd); m = m.to(DEV)
            st = ShardedPretrainStep(m, args, N, DistNodeGroup(0, 1), synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=True)
            out = []
            for i, (src, lc) in enumerate(zip(srcs, orders)):
                if lose and i == 1:
                    torch.cuda.synchronize()
                    _C.lib().call("gptst_handoff_inject", 1)
                st.step(src, 20, list_c=lc)
                out.append(st.losses())
            assert st.safe_mode == lose and st.lost_steps == (1 if lose else 0) and (st.tA, st.tB) == (3, 3)
            res.append((out, m.flat.detach().clone()))
    finally:
        _C.lib().call("gptst_handoff_reset")
    for a, b in zip(res[0][0], res[1][0]):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-4 * max(abs(y), 1e-3), (res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).norm() / res[0][1].norm()) < 1e-3
