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


def _args(n):
    return make_args("PEMS08", num_nodes=n, embed_dim=8, HS=5, HT=6, num_route=2, scaler_zeros=synth.scaler_zeros(), epochs=30,
                     change_epoch=3)


@pytest.mark.parametrize("W", [2, 4])
def test_node_shards_equal_unsharded_step(W):
    from gptst_amd import ops
    from gptst_amd.model import GPTST_Model
    from gptst_amd.shard import ShardedPretrainStep, ThreadNodeGroup, shard_state_dict, unshard_state_dicts
    from gptst_amd.step import PretrainStep
    N, B = 40, 2
    Nl = N // W
    args_g = _args(N)
    sd = O.init_state_dict(args_g, 5)
    Mg = B * 12 * N
    steps = [(1, 0), (20, 1), (25, 2)]                     # (epoch, seed): random phase, then adaptive + KL twice
    srcs = [synth.make_batch(B, 12, N, 1, seed=40 + s).to(DEV) for _, s in steps]
    noise = [tuple(synth.make_noise(Mg, 10 * s + i).to(DEV) for i in range(3)) for _, s in steps]
    list_c = [3, 1, 0, 4, 2]

    # ---- unsharded reference ----
    model = GPTST_Model(args_g); model.load_state_dict(sd); model = model.to(DEV)
    st = PretrainStep(model, args_g, synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B, use_graph=False)
    ref_loss, ref_mask = [], []
    for (epoch, _), src, (n0, na, nr) in zip(steps, srcs, noise):
        st.step(src, epoch, noise=n0, noise_a=na, noise_r=nr, list_c=list_c)
        ref_loss.append(st.losses()); ref_mask.append(st.last_mask.clone())
    ref_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    # ---- two shards, one thread each ----
    shared = ThreadNodeGroup.Shared(W)
    ops.CALL_LOCK = threading.Lock()
    out, errs = [None] * W, []

    def rank_main(r):
        try:
            args_l = _args(Nl)
            m = GPTST_Model(args_l); m.load_state_dict(shard_state_dict(sd, r * Nl, (r + 1) * Nl)); m = m.to(DEV)
            s = ShardedPretrainStep(m, args_l, N, ThreadNodeGroup(r, shared), synth.SCALER_MEAN, synth.SCALER_STD, batch_size=B)
            losses, masks = [], []
            for (epoch, _), src, (n0, na, nr) in zip(steps, srcs, noise):
                s.step(src[:, :, r * Nl:(r + 1) * Nl].contiguous(), epoch, noise=n0, noise_a=na, noise_r=nr, list_c=list_c)
                losses.append(s.losses()); masks.append(s.last_mask_global.clone())
            out[r] = (losses, masks, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
        except BaseException as e:              # noqa: BLE001 - surface the failure in the main thread
            errs.append(e)
            shared.barrier.abort()

    try:
        ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(300)
    finally:
        ops.CALL_LOCK = None
    assert not errs, errs
    got_sd = unshard_state_dicts([out[r][2] for r in range(W)])
    for i in range(len(steps)):
        for r in range(W):
            assert torch.equal(out[r][1][i], ref_mask[i]), "global mask differs at step %d on rank %d" % (i, r)
            for a, b in zip(out[r][0][i], ref_loss[i]):
                assert abs(a - b) <= 2e-4 * max(abs(b), 1e-3), (i, r, out[r][0][i], ref_loss[i])
    worst = 0.0
    for k, v in ref_sd.items():
        if not v.dtype.is_floating_point:
            continue
        upd = v - sd[k]
        err = float((got_sd[k] - v).norm() / upd.norm().clamp_min(1e-6))
        worst = max(worst, err)
        assert err < 2e-3, "%s: update differs, rel-L2 of the update error %.3e" % (k, err)
    print("worst relative update error %.2e" % worst)
